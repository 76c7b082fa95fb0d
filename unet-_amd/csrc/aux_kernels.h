// aux_kernels.h — the non-GEMM kernels of the UNet++ hot path (all HBM-bound, 16-byte lanes):
//   weight repack (canonical fp32 OIHW -> per-tile fp16 hi/lo slabs), input conversion,
//   bilinear-2x upsample (the concat is virtual), 1x1 head + argmax + class masks, debug unpack.
#pragma once
#include "conv3x3_mfma.h"

namespace unetpp {

// ------------------------------------------------------------------------------------------------
// Per-output-channel power-of-two weight scaling: scaled weights have max |w| in [2^13, 2^14), so
// the fp16 `lo` plane of a split weight stays in the normal range (22 significant bits overall).
// scale_out[co] = 2^-k is applied to the fp32 accumulator in the conv epilogue (exact).
// A non-finite weight or bias (the only way a NaN can get into an accumulator) sets ST_NAN in the engine's status.
__global__ void weight_scale_kernel(const float* __restrict__ w, int per_co, const float* __restrict__ bias,
                                    float* __restrict__ mult_out, float* __restrict__ scale_out, unsigned* __restrict__ status) {
  const int co = blockIdx.x;
  float m = 0.f;
  bool bad = threadIdx.x == 0 && !(fabsf(bias[co]) <= 3.0e38f);
  for (int i = threadIdx.x; i < per_co; i += blockDim.x) {
    const float x = fabsf(w[(size_t)co * per_co + i]);
    bad |= !(x <= 3.0e38f);
    m = fmaxf(m, x);
  }
  if (bad) atomicOr(status, ST_NAN);
  __shared__ float red[256];
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int e = 0;
    float mm = red[0];
    int k = 0;
    if (mm > 0.f && mm < 3.0e38f) {
      frexpf(mm, &e);  // mm = f * 2^e, f in [0.5,1)
      k = 14 - e;
    }
    mult_out[co] = ldexpf(1.0f, k);
    scale_out[co] = ldexpf(1.0f, -k);
  }
}

// canonical OIHW fp32 -> [ct][chunk][P][tap][KG][BN][8] fp16 (plane 0 = hi, plane 1 = lo).
// rows8 (layers of the wave-specialised kernel): inside every 32-row block of the MFMA's A operand, row
// (r & 3) + 8 (r >> 2) + 4 h carries output channel 16 (r >> 3) + 8 h + (r & 7) -- the accumulator layout of
// v_mfma_f32_32x32x16 then leaves a lane (h = lane >> 5) with channels 8 h .. 8 h + 7 of the tile's first 16-channel
// record in registers 0..7 and the same eight of its second record in registers 8..15: the epilogue stores 16-byte
// pieces without exchanging words between the two half-waves (8 v_permlane32_swap per row otherwise, 12-16 cycles each).
__global__ void weight_pack_kernel(const float* __restrict__ w, const float* __restrict__ mult, int Cin, int Cout,
                                   int P, int KC, int BN, int nchunks, half_t* __restrict__ out, long long units, int rows8) {
  long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const int KG = KC / 8;
  long long t = u;
  int n = t % BN; t /= BN;
  int kg = t % KG; t /= KG;
  int tap = t % 9; t /= 9;
  int pl = t % P; t /= P;
  int c = t % nchunks; t /= nchunks;
  int ct = (int)t;
  int co = ct * BN + n;
  if (rows8) {
    const int nb = n & 31, h = (nb >> 2) & 1, r = (nb & 3) + 4 * (nb >> 3);
    co = ct * BN + (n & ~31) + 16 * (r >> 3) + 8 * h + (r & 7);
  }
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    int ci = c * KC + kg * 8 + e;
    float v = 0.f;
    if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * 9 + tap] * mult[co];
    half_t hi = (half_t)v;
    r[e] = pl == 0 ? hi : (half_t)(v - (float)hi);
  }
  *(half8*)(out + u * 8) = r;
}

// EXACT8 slabs of the wave-specialised kernel (conv3x3_ws.h, X8): per (channel tile, chunk of 16 input channels)
//   main   [tap 9][k-group 2][BN][8 halves]         fp16(w 2^k), the hi plane of weight_pack_kernel (rows8 order)
//   cross  [pair 5][tap of the pair 2][h 2][BN][16 bytes]   e4m3 bytes; lane (n, h) of the scaled MFMA's first operand holds,
//          per tap, [wh8 x 4 | wl8 x 4 | wh8 x 4 | wl8 x 4] for input channels 8 h + 0..3 and 8 h + 4..7 -- the order of the
//          activations' [lo8 x 4 | x8 x 4 | ...] bytes (conv3x3_mfma.h) -- with
//             wh8 = e4m3(2^-6 ws)   (ws = w 2^k, max |ws| in [2^13, 2^14): at most 256)
//             wl8 = e4m3(2^5 (ws - fp16(ws)))   (|ws - fp16(ws)| <= 4: at most 128)
//          so that with the block scales 2^6 (weights) and 2^-8 (activations: lo8 = 2^8 lo, x8 = 2^-3 x) the instruction adds
//          ws * lo + (ws - fp16(ws)) * x.   Pairs (taps as dy * 3 + dx): (0,1) (3,4) (6,7) (2,5) (-,8): the consumers' order.
// pair9: the fifth pair of an ODD chunk holds tap 8 of the chunk before it and tap 8 of its own (the consumers keep the even
// chunk's operand in registers until then); the fifth pair of an even chunk is not used.
__global__ void weight_pack_x8_kernel(const float* __restrict__ w, const float* __restrict__ mult, int Cin, int Cout, int BN, int nchunks,
                                      char* __restrict__ out, long long units, int pair9) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  const int per = 38 * BN;
  const long long t = u / per;
  int r = (int)(u - t * per);
  const int c = (int)(t % nchunks), ct = (int)(t / nchunks);
  const bool is_main = r < 18 * BN;
  if (!is_main) r -= 18 * BN;
  const int n = r % BN;
  const int nb = n & 31, hh = (nb >> 2) & 1, rr = (nb & 3) + 4 * (nb >> 3);
  const int co = ct * BN + (n & ~31) + 16 * (rr >> 3) + 8 * hh + (rr & 7);        // rows8, see weight_pack_kernel
  const float mu = mult[co];
  if (is_main) {
    const int kg = (r / BN) & 1, tap = r / (2 * BN);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int ci = c * 16 + kg * 8 + e;
      o[e] = (half_t)(ci < Cin ? w[((size_t)co * Cin + ci) * 9 + tap] * mu : 0.f);
    }
    *(half8*)(out + u * 16) = o;
    return;
  }
  const int h = (r / BN) & 1, half = (r / (2 * BN)) & 1, pair = r / (4 * BN);
  // pair 4: bytes 16-31 = tap 8 of this chunk; bytes 0-15 = tap 8 of the chunk before (pair9, odd chunks) or nothing
  int tap = pair < 3 ? pair * 3 + half : (pair == 3 ? (half ? 5 : 2) : (half ? 8 : -1));
  int cc = c;                     // chunk whose channels this 16-byte unit holds
  if (pair9 && pair == 4) {
    if (c & 1) { tap = 8; cc = half ? c : c - 1; } else tap = -1;
  }
  unsigned wd[4] = {0u, 0u, 0u, 0u};
  if (tap >= 0) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float hi4[4], lo4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ci = cc * 16 + 8 * h + 4 * q + i;
        const float ws = ci < Cin ? w[((size_t)co * Cin + ci) * 9 + tap] * mu : 0.f;
        hi4[i] = ws * 0.015625f;
        lo4[i] = (ws - (float)(half_t)ws) * 32.0f;
      }
      int a = 0, b = 0;
      a = __builtin_amdgcn_cvt_pk_fp8_f32(hi4[0], hi4[1], a, false); a = __builtin_amdgcn_cvt_pk_fp8_f32(hi4[2], hi4[3], a, true);
      b = __builtin_amdgcn_cvt_pk_fp8_f32(lo4[0], lo4[1], b, false); b = __builtin_amdgcn_cvt_pk_fp8_f32(lo4[2], lo4[3], b, true);
      wd[2 * q] = (unsigned)a; wd[2 * q + 1] = (unsigned)b;
    }
  }
  *(u32x4*)(out + u * 16) = (u32x4){wd[0], wd[1], wd[2], wd[3]};
}

// conv0_0.conv1 weights (canonical OIHW fp32 [32][3][3][3]) -> A fragments of v_mfma_f32_16x16x32_f16 for the fused first
// block (conv3x3_ws.h, C0F): [half c][plane][lane][8]; lane l holds output channel 16 c + (l & 15), operand slots
// 8 (l >> 4) .. +7 of the K = 32 order [dy 0: (dx,ch) 0..7][dy 1: 0..7][dy 2: 0..7][(dx,ch) = 8 of dy 0, 1, 2][0 x 5].
__global__ void conv0_pack_kernel(const float* __restrict__ w, const float* __restrict__ mult, half_t* __restrict__ out) {
  const int u = blockIdx.x * blockDim.x + threadIdx.x;      // (c, plane, lane)
  if (u >= 2 * 2 * 64) return;
  const int lane = u & 63, pl = (u >> 6) & 1, c = u >> 7;
  const int co = 16 * c + (lane & 15), kq = lane >> 4;
  half8 r;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int dy = -1, j = 0;
    if (kq < 3) { dy = kq; j = i; } else if (i < 3) { dy = i; j = 8; }
    float v = 0.f;
    if (dy >= 0) { const int dx = j / 3, ch = j - 3 * dx; v = w[((co * 3 + ch) * 3 + dy) * 3 + dx] * mult[co]; }
    const half_t hi = (half_t)v;
    r[i] = pl == 0 ? hi : (half_t)(v - (float)hi);
  }
  *(half8*)(out + (size_t)u * 8) = r;
}

// ------------------------------------------------------------------------------------------------
// Input conversion -> [N][1][H][W][P][8] fp16 (one channel block of 8, channels 3..7 zero).
//   fmt 0: float32 NCHW RGB in [0,1]                    (model(img_tensor), infer_two_stage_burr.py:292-295)
//   fmt 1: uint8 NHWC BGR: RGB = BGR reversed, /255.0f   (preprocess_image, infer_two_stage_burr.py:122-127)
template <int P, bool X8 = false>
__global__ void convert_input_kernel(const void* __restrict__ in, int fmt, int N, int H, int W,
                                     half_t* __restrict__ out, unsigned* __restrict__ status) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * H * W;
  if (i >= total) return;
  float v[3];
  if (fmt == 0) {
    size_t hw = (size_t)H * W;
    size_t n = i / hw, p = i - n * hw;
    const float* f = (const float*)in + n * 3 * hw + p;
    v[0] = f[0]; v[1] = f[hw]; v[2] = f[2 * hw];
  } else {
    const uint8_t* b = (const uint8_t*)in + i * 3;
    v[0] = __fdiv_rn((float)b[2], 255.0f);
    v[1] = __fdiv_rn((float)b[1], 255.0f);
    v[2] = __fdiv_rn((float)b[0], 255.0f);
  }
  // a float32 input outside the fp16 range (or a NaN) cannot enter the hi/lo planes unchanged: report it
  const bool beyond = !(fabsf(v[0]) <= F16_MAX) || !(fabsf(v[1]) <= F16_MAX) || !(fabsf(v[2]) <= F16_MAX);
  if (__builtin_amdgcn_ballot_w64(beyond)) {
    const bool nan = v[0] != v[0] || v[1] != v[1] || v[2] != v[2];
    const unsigned long long bn = __builtin_amdgcn_ballot_w64(nan), bo = __builtin_amdgcn_ballot_w64(beyond && !nan);
    if (beyond && status) atomicOr(status, (bn ? ST_NAN : 0u) | (bo ? ST_OVERFLOW : 0u));
  }
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hi[e] = (half_t)0.f; lo[e] = (half_t)0.f; }
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    v[e] = __builtin_amdgcn_fmed3f(v[e], -F16_MAX, F16_MAX);
    half_t h = (half_t)v[e];
    hi[e] = h;
    lo[e] = (half_t)(v[e] - (float)h);
  }
  half8* o = (half8*)(out + i * P * 8);
  o[0] = hi;
  if (X8) {      // EXACT8: the second 16 bytes are {lo8 x 4, x8 x 4} of channels 0..3 (the blue slot and channels 4..7 are zero)
    unsigned h0, h1, l8, x8;
    split_pack4_x8(v[0], v[1], v[2], 0.f, h0, h1, l8, x8);
    *(u32x4*)(o + 1) = (u32x4){l8, x8, 0u, 0u};
  } else if (P == 2) o[1] = lo;
}

// ------------------------------------------------------------------------------------------------
// up = nn.Upsample(scale_factor=2, 'bilinear', align_corners=True) (reference unetpp.py:76,112-116).
// src = dst*(in-1)/(out-1) in fp32, i0 = int(src), i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1;
// x is interpolated inside each row first.  The torch.cat([skip, up]) that follows is virtual: the
// decoder conv reads `skip` and this kernel's output as two sources (conv3x3_mfma.h).
typedef __attribute__((ext_vector_type(4))) _Float16 half4v;
constexpr int UP_SEG = 256;                 // output columns per upsample workgroup

template <int P>
__global__ __launch_bounds__(1024) void upsample2x_kernel(const half_t* __restrict__ low, int H, int W,
                                                         half_t* __restrict__ out) {
  // Tensors are channel-blocked: [N][Cu/16][h][w][P][16]; a pixel record is 2*P pieces of 16 bytes
  // ([hi 0-7][hi 8-15][lo 0-7][lo 8-15] in EXACT mode).
  // grid = (H/2 * ceil(W/UP_SEG), N * Cu/16): one workgroup produces UP_SEG columns of output rows 2r, 2r+1 of one
  // channel block.  The (at
  // most three) low-res rows they interpolate are staged ONCE in LDS with full-line loads, so the four
  // corner reads per output are LDS reads: the vector-memory pipe only sees ~1.75 instructions per output
  // KiB instead of 10 (the register version was bound by the address coalescer, not by HBM).
  // One thread stores one 16-byte piece of each of the two rows, so a wave's store is 1 KiB of consecutive bytes; in
  // EXACT mode the two threads that share a channel octet (lane ^ 2) each interpolate four of its eight channels
  // (hi+lo in fp32) and swap halves with one DPP exchange.
  // Workgroups r, r+8, ... share an XCD (round-robin dispatch): the row-pair index is permuted so that each
  // XCD owns a contiguous band of rows and neighbouring row pairs find their shared low-res row in its L2.
  constexpr int PIECES = 2 * P;
  constexpr int REC = P * 16;                 // halves per pixel record
  extern __shared__ __attribute__((aligned(16))) char up_smem[];
  half_t* rows = (half_t*)up_smem;            // [3][wseg][REC]
  const int HP = H >> 1;
  // a row pair is cut into column segments of SEG outputs so that the staged rows stay <= 25 KB per
  // workgroup (enough resident workgroups per CU to cover the staging loads)
  const int nseg = (W + UP_SEG - 1) / UP_SEG;
  const int rb = blockIdx.x / nseg, seg = blockIdx.x - rb * nseg;
  const int r = (HP % 8 == 0) ? (rb & 7) * (HP >> 3) + (rb >> 3) : rb;
  const size_t nb = blockIdx.y;               // n * (Cu/16) + channel block
  const int h = H >> 1, w = W >> 1;
  const float sh = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  const int ybase = (int)(sh * (float)(2 * r));          // first low-res row needed
  const int xo0 = seg * UP_SEG, xo1 = min(W, xo0 + UP_SEG);                    // output columns of this segment
  const int xbase = (int)(sw * (float)xo0);                                     // first low-res column needed
  const int wseg = min(w - 1, (int)(sw * (float)(xo1 - 1)) + 1) - xbase + 1;    // low-res columns staged
  {   // stage low-res rows ybase .. ybase+2 (clamped), columns xbase .. xbase+wseg-1: 16 bytes per thread and step
    const half_t* src = low + nb * (size_t)h * w * REC;
    const int units_row = wseg * REC / 8;
    for (int u = threadIdx.x; u < 3 * units_row; u += blockDim.x) {
      const int rr = u / units_row, c = u - rr * units_row;
      const int yy = min(ybase + rr, h - 1);
      *(u32x4*)(rows + (size_t)u * 8) = *(const u32x4*)(src + ((size_t)yy * w + xbase) * REC + (size_t)c * 8);
    }
  }
  __syncthreads();
  // ---- one thread = one 16-byte piece of one output column, for BOTH output rows: the x-interpolation of the staged
  // low-res rows (the expensive part: fp16 hi+lo -> fp32) is done once and shared; the rows differ only in their
  // y-weights, folded into one weight per staged row (zero for a row that output row does not touch).
  float wy[2][3];
#pragma unroll
  for (int ro = 0; ro < 2; ++ro) {
    const int y = 2 * r + ro;
    const float fy = sh * (float)y;
    const int y0 = (int)fy;
    const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), ly0 = 1.f - ly1;
#pragma unroll
    for (int k = 0; k < 3; ++k) wy[ro][k] = (k == y0 - ybase ? ly0 : 0.f) + (k == y1 - ybase ? ly1 : 0.f);
  }
  const int row_items = (xo1 - xo0) * PIECES;  // multiple of 64: every wave is full (DPP exchange below)
  for (int idx = threadIdx.x; idx < row_items; idx += blockDim.x) {
    const int piece = idx & (PIECES - 1);
    const int x = xo0 + idx / PIECES;
    const float fx = sw * (float)x;
    const int xg = (int)fx;                     // global low-res column
    const int x0 = xg - xbase;                  // column inside the staged segment
    const int x1 = x0 + (xg < w - 1 ? 1 : 0);
    const float lx1 = fminf(fmaxf(fx - (float)xg, 0.f), 1.f), lx0 = 1.f - lx1;
    half_t* dst = out + ((nb * H + 2 * r) * W + x) * (size_t)REC + piece * 8;      // output row 2r; row 2r+1 is W*REC further
    if (P == 2) {
      const int oct = piece & 1, role = piece >> 1;       // role 0 stores hi and computes channels 0-3 of the octet
      const int e0 = oct * 8 + role * 4;
      float t[3][4];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const half_t* rk = rows + (size_t)k * wseg * REC + e0;
        const half4v a0 = *(const half4v*)(rk + x0 * REC), a1 = *(const half4v*)(rk + x1 * REC);
        const half4v b0 = *(const half4v*)(rk + x0 * REC + 16), b1 = *(const half4v*)(rk + x1 * REC + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e)       // lx0 * (hi + lo) + lx1 * (hi + lo) as four mixed-precision FMAs
          t[k][e] = fmaf(lx1, (float)b1[e], fmaf(lx1, (float)a1[e], fmaf(lx0, (float)b0[e], lx0 * (float)a0[e])));
      }
#pragma unroll
      for (int ro = 0; ro < 2; ++ro) {
        half4v rh, rl;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = fmaf(wy[ro][2], t[2][e], fmaf(wy[ro][1], t[1][e], wy[ro][0] * t[0][e]));
          half_t hi, lo;
          split_f16(v, hi, lo);
          rh[e] = hi; rl[e] = lo;
        }
        typedef __attribute__((ext_vector_type(2))) int int2v;
        const int2v keep = __builtin_bit_cast(int2v, role ? rl : rh);
        const int2v send = __builtin_bit_cast(int2v, role ? rh : rl);
        int2v recv;
        recv[0] = __builtin_amdgcn_mov_dpp(send[0], 0x4E, 0xF, 0xF, true);     // quad_perm [2,3,0,1]: lane ^ 2
        recv[1] = __builtin_amdgcn_mov_dpp(send[1], 0x4E, 0xF, 0xF, true);
        // hi lane: [own channels 0-3 | partner's 4-7]; lo lane: [partner's 0-3 | own 4-7]
        u32x4 o = role ? u32x4{(unsigned)recv[0], (unsigned)recv[1], (unsigned)keep[0], (unsigned)keep[1]}
                       : u32x4{(unsigned)keep[0], (unsigned)keep[1], (unsigned)recv[0], (unsigned)recv[1]};
        // written once, next read after >1 GB of other traffic
        __builtin_nontemporal_store(o, (u32x4*)(dst + (size_t)ro * W * REC));
      }
    } else {
      const int e0 = piece * 8;
      float t[3][8];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const half_t* rk = rows + (size_t)k * wseg * REC + e0;
        const half8 a0 = *(const half8*)(rk + x0 * REC), a1 = *(const half8*)(rk + x1 * REC);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[k][e] = fmaf(lx1, (float)a1[e], lx0 * (float)a0[e]);
      }
#pragma unroll
      for (int ro = 0; ro < 2; ++ro) {
        half8 rr;
#pragma unroll
        for (int e = 0; e < 8; ++e)
          rr[e] = (half_t)fmaf(wy[ro][2], t[2][e], fmaf(wy[ro][1], t[1][e], wy[ro][0] * t[0][e]));
        __builtin_nontemporal_store(rr, (half8*)(dst + (size_t)ro * W * REC));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone 1x1 head: self.final = Conv2d(Cx, C, 1) in fp32 (NestedUNet unetpp.py:85,119 with Cx = 32 —
// only in debug mode, normally fused into conv0_4.conv2 — and SimpleUNet simple_unet.py:92,124 with
// Cx = 64), then softmax probabilities, argmax (first maximal class, np.argmax) and the class rules
// (apply_rule, conv3x3_mfma.h).  x is channel-blocked [N][Cx/16][H][W][P][16]; C <= HEAD_FUSED_MAX_CLASSES.
template <int P>
__global__ void head_generic_kernel(const half_t* __restrict__ x, int Cx, const float* __restrict__ w /*[C][Cx]*/,
                                    const float* __restrict__ b, int C, int H, int W,
                                    float* __restrict__ logits, float* __restrict__ probs, uint8_t* __restrict__ mask,
                                    uint8_t* __restrict__ cable, uint8_t* __restrict__ tape, int rule, float t_cable,
                                    float t_tape, float bg_margin, float ct_margin) {
  extern __shared__ float head_ws[];          // [C][Cx] weights then [C] biases
  for (int i = threadIdx.x; i < C * Cx; i += blockDim.x) head_ws[i] = w[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) head_ws[C * Cx + i] = b[i];
  __syncthreads();
  const unsigned hw = (unsigned)(H * W);
  const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;     // pixel inside image blockIdx.y
  const unsigned n = blockIdx.y;
  if (p >= hw) return;
  float lg[HEAD_FUSED_MAX_CLASSES];
#pragma unroll
  for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) lg[c] = (c < C) ? head_ws[C * Cx + c] : -INFINITY;
  const int nblk = Cx >> 4;
  for (int blk = 0; blk < nblk; ++blk) {
    const half8* px = (const half8*)(x + ((((size_t)n * nblk + blk) * hw + p) * (P == 3 ? 2 : P)) * 16);
    float v[16];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      half8 hi = px[g];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[g * 8 + e] = (float)hi[e];
      if (P == 3) {      // EXACT8 record: value = hi + 2^-8 lo8, lo8 bytes of channels 8 g + 4 q + i in word 2 q of piece 2 + g
        const u32x4 w8 = __builtin_bit_cast(u32x4, px[2 + g]);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          v[g * 8 + 4 * q + 0] += __builtin_amdgcn_cvt_scalef32_f32_bf8((int)w8[2 * q], X8_LO_MUL, 0);
          v[g * 8 + 4 * q + 1] += __builtin_amdgcn_cvt_scalef32_f32_bf8((int)w8[2 * q], X8_LO_MUL, 1);
          v[g * 8 + 4 * q + 2] += __builtin_amdgcn_cvt_scalef32_f32_bf8((int)w8[2 * q], X8_LO_MUL, 2);
          v[g * 8 + 4 * q + 3] += __builtin_amdgcn_cvt_scalef32_f32_bf8((int)w8[2 * q], X8_LO_MUL, 3);
        }
      }
      if (P == 2) {
        half8 lo = px[2 + g];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[g * 8 + e] += (float)lo[e];
      }
    }
#pragma unroll
    for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c)
      if (c < C) {
        float s = lg[c];
#pragma unroll
        for (int e = 0; e < 16; ++e) s = fmaf(v[e], head_ws[c * Cx + blk * 16 + e], s);
        lg[c] = s;
      }
  }
  float best = lg[0];
  int besti = 0;
#pragma unroll
  for (int c = 1; c < HEAD_FUSED_MAX_CLASSES; ++c)
    if (lg[c] > best) { best = lg[c]; besti = c; }
  bool is_cable = besti == 1, is_tape = besti == 2;
  const size_t o = (size_t)n * hw + p;
  if (probs || rule) {
    float pe[HEAD_FUSED_MAX_CLASSES], sum = 0.f;
#pragma unroll
    for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) { pe[c] = (c < C) ? expf(lg[c] - best) : 0.f; sum += pe[c]; }
#pragma unroll
    for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) {
      pe[c] = pe[c] / sum;
      if (probs && c < C) probs[((size_t)n * C + c) * hw + p] = pe[c];
    }
    if (rule) apply_rule(rule, pe[0], pe[1], pe[2], t_cable, t_tape, bg_margin, ct_margin, is_cable, is_tape);
  }
  if (logits) {
#pragma unroll
    for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c)
      if (c < C) logits[((size_t)n * C + c) * hw + p] = lg[c];
  }
  if (mask) mask[o] = (uint8_t)besti;
  if (cable) cable[o] = is_cable;
  if (tape) tape[o] = is_tape;
}

// ---- frame glue: cv2.resize either side of the model ----------------------------------
// Both kernels take per-axis tables the host builds the way OpenCV's resizeGeneric_/resizeNN do (double/float
// index math, 11-bit fixed-point coefficients), so the device side is integer-only.
//   lin table entry {s0, s1, a0, a1}: out = s0-th and s1-th source sample weighted a0, a1 (a0 + a1 = 2048)
// resize_linear_u8_kernel: cv2.resize(frame, (dw, dh), INTER_LINEAR) for B interleaved C-channel uint8 frames
// (infer_two_stage_burr.py:124).  Horizontal pass in int32, vertical pass (((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
// grid = (ceil(dw*C / (4*256)), dh, B); a thread produces 4 consecutive output bytes.
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw, int C,
                                                               uint8_t* __restrict__ dst, int dh, int dw,
                                                               const int4* __restrict__ xtab, const int4* __restrict__ ytab) {
  const int row_bytes = dw * C;
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= row_bytes) return;
  const int dy = blockIdx.y;
  const int4 yt = ytab[dy];
  const uint8_t* r0 = src + ((size_t)blockIdx.z * sh + yt.x) * (size_t)sw * C;
  const uint8_t* r1 = src + ((size_t)blockIdx.z * sh + yt.y) * (size_t)sw * C;
  uint8_t* o = dst + ((size_t)blockIdx.z * dh + dy) * (size_t)row_bytes + i0;
  uint32_t packed = 0;
  const int n = min(4, row_bytes - i0);
  for (int j = 0; j < n; ++j) {
    const int i = i0 + j;
    const int dx = i / C, c = i - dx * C;
    const int4 xt = xtab[dx];
    const int S0 = (int)r0[xt.x * C + c] * xt.z + (int)r0[xt.y * C + c] * xt.w;
    const int S1 = (int)r1[xt.x * C + c] * xt.z + (int)r1[xt.y * C + c] * xt.w;
    int v = (((yt.z * (S0 >> 4)) >> 16) + ((yt.w * (S1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
    packed |= (uint32_t)v << (8 * j);
  }
  if (n == 4 && (row_bytes & 3) == 0) *(uint32_t*)o = packed;
  else for (int j = 0; j < n; ++j) o[j] = (uint8_t)(packed >> (8 * j));
}

// resize_nearest_roi_u8_kernel: (pred == match_class) [or the mask itself when match_class < 0], cv2.resize(...,
// (dw, dh), INTER_NEAREST), then zero outside rows [y1, y2) x columns [x1, x2) — infer_two_stage_burr.py:303-314.
// xofs/yofs: source index per destination column/row.  grid = (ceil(dw / (4*256)), dh, B).
__global__ __launch_bounds__(256) void resize_nearest_roi_u8_kernel(const uint8_t* __restrict__ src, int sh, int sw,
                                                                    uint8_t* __restrict__ dst, int dh, int dw,
                                                                    const int* __restrict__ xofs, const int* __restrict__ yofs,
                                                                    int match_class, int x1, int y1, int x2, int y2) {
  const int i0 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= dw) return;
  const int dy = blockIdx.y;
  const bool row_in = dy >= y1 && dy < y2;
  const uint8_t* r = src + ((size_t)blockIdx.z * sh + yofs[dy]) * (size_t)sw;
  uint8_t* o = dst + ((size_t)blockIdx.z * dh + dy) * (size_t)dw + i0;
  uint32_t packed = 0;
  const int n = min(4, dw - i0);
  for (int j = 0; j < n; ++j) {
    const int dx = i0 + j;
    uint32_t v = 0;
    if (row_in && dx >= x1 && dx < x2) {
      v = r[xofs[dx]];
      if (match_class >= 0) v = v == (uint32_t)match_class ? 1u : 0u;
    }
    packed |= v << (8 * j);
  }
  if (n == 4 && (dw & 3) == 0) *(uint32_t*)o = packed;
  else for (int j = 0; j < n; ++j) o[j] = (uint8_t)(packed >> (8 * j));
}


// ------------------------------------------------------------------------------------------------
// Per-frame mask statistics so that the uint8 mask need not leave the GPU when the
// host only wants counts and widths:
//   counts[b][c]         = number of pixels of class c            (np.sum(mask_cable), infer_two_stage_burr.py:333-334;
//                                                                  cable_coverage = sum / (H*W), geometry_enhanced.py:151-152)
//   row_min/max[b][c][y] = first / last column of class c in row y (W / -1 when the row has none): the per-row
//                          width xs.max() - xs.min() + 1 of _compute_width_per_row (geometry_enhanced.py:45-74)
// grid = (H, B), one workgroup per mask row; classes >= C are ignored.
__global__ __launch_bounds__(256) void mask_stats_kernel(const uint8_t* __restrict__ mask, int C, int H, int W,
                                                         unsigned* __restrict__ counts, int* __restrict__ row_min,
                                                         int* __restrict__ row_max) {
  __shared__ int s_min[HEAD_MAX_CLASSES], s_max[HEAD_MAX_CLASSES];
  __shared__ unsigned s_cnt[HEAD_MAX_CLASSES];
  const int y = blockIdx.x, b = blockIdx.y;
  if ((int)threadIdx.x < C) { s_min[threadIdx.x] = W; s_max[threadIdx.x] = -1; s_cnt[threadIdx.x] = 0; }
  __syncthreads();
  const uint8_t* row = mask + ((size_t)b * H + y) * W;
  for (int x = threadIdx.x; x < W; x += blockDim.x) {
    const int c = row[x];
    if (c < C) {
      atomicMin(&s_min[c], x);
      atomicMax(&s_max[c], x);
      atomicAdd(&s_cnt[c], 1u);
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < C) {
    const int c = threadIdx.x;
    row_min[((size_t)b * C + c) * H + y] = s_min[c];
    row_max[((size_t)b * C + c) * H + y] = s_max[c];
    if (s_cnt[c]) atomicAdd(&counts[(size_t)b * C + c], s_cnt[c]);
  }
}

// debug: channel-blocked planes -> float32 NCHW.  plane 0: the value a reader reconstructs (hi [+ lo | + 2^-8 lo8]);
// 1: the hi plane alone; 2: the second plane as stored (fp16 lo, or the decoded e5m2 lo8 = e5m2(2^8 lo)); 3: EXACT8's decoded
// x8 = e5m2(2^-3 v) -- the layer-by-layer arithmetic test (tests/test_gpu_exact8.py) needs the planes themselves.
template <int P>
__global__ void unpack_nchw_kernel(const half_t* __restrict__ x, int N, int C, int H, int W, float* __restrict__ out, int plane) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * C * H * W;
  if (i >= total) return;
  int xx = i % W; size_t t = i / W;
  int y = t % H; t /= H;
  int c = t % C; int n = t / C;
  const int CB = C < 16 ? C : 16;                 // channel-blocked source: [N][C/CB][H][W][P][CB]
  const half_t* p = x + ((((size_t)(n * (C / CB) + c / CB) * H + y) * W + xx) * (P == 3 ? 2 : P)) * CB + c % CB;
  float v = (float)p[0];
  if (P == 2) {
    const float lo = (float)p[CB];
    v = plane == 0 ? v + lo : plane == 2 ? lo : v;
  }
  if (P == 3) {          // EXACT8 record (CB = 16): lo8 at byte 32 + 16 (c / 8) + 8 ((c / 4) & 1) + c % 4, x8 four bytes behind it
    const int cc = c % 16;
    const unsigned char* q = (const unsigned char*)(p - cc) + 32 + 16 * (cc >> 3) + 8 * ((cc >> 2) & 1) + (cc & 3);
    const float l8 = __builtin_amdgcn_cvt_scalef32_f32_bf8((int)q[0], 1.0f, 0), x8 = __builtin_amdgcn_cvt_scalef32_f32_bf8((int)q[4], 1.0f, 0);
    v = plane == 0 ? v + l8 * X8_LO_MUL : plane == 2 ? l8 : plane == 3 ? x8 : v;
  }
  out[i] = v;
}

}  // namespace unetpp
