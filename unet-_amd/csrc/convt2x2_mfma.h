// convt2x2_mfma.h — nn.ConvTranspose2d(Cin, Cout, kernel_size=2, stride=2) (reference
// src/models/simple_unet.py:62-64, used at :109,114,119) as an MFMA GEMM:
//   out[n][co][2y+dy][2x+dx] = b[co] + sum_ci in[n][ci][y][x] * W[ci][co][dy][dx]
// = a 1x1 convolution with 4*Cout "virtual channels" v = (dy*2+dx)*Cout + co followed by a pixel shuffle
// that the epilogue does while storing.  It is ~4 % of SimpleUNet's FLOPs, so the kernel is deliberately
// simple: no LDS, no barriers; every wave owns a 128-pixel x 64-virtual-channel tile (8 accumulators)
// and streams its MFMA operands straight from global memory / L2 (12 x 1 KiB loads per 24 MFMAs, register
// double-buffered one K-step ahead).
// Same operand conventions as conv3x3_mfma.h: weights are the A operand (virtual channel on the row),
// pixels the B operand (pixel on the lane); EXACT mode (P = 2) issues lo*hi, hi*lo, hi*hi.
#pragma once
#include "conv3x3_mfma.h"

namespace unetpp {

struct ConvTArgs {
  const half_t* in;      // [N][Cin/16][H][W][P][16]
  const half_t* wpk;     // packed weights [v-tile (32)][Cin/16][P][64 lanes][8]: the A fragment of lane l
  const float* scale;    // [Cout] 2^-k undoing the per-channel weight scaling
  const float* bias;     // [Cout]
  half_t* out;           // [N][Cout/16][2H][2W][P][16]
  int N, H, W, Cin, Cout;
  unsigned* status;      // engine's sticky range flags (conv3x3_mfma.h: range_flag)
};

// X8 (precision exact8): the input's second plane holds {lo8 x 4, x8 x 4, lo8 x 4, x8 x 4} per channel octet instead of fp16 lo;
// the residual is decoded back to fp16 (2^-8 e5m2: exact) and the three fp16 terms are issued as in exact mode -- 4 % of the
// network's flops, not worth a scaled-MFMA path of its own; the output is stored with the 8-bit planes.
template <int P, bool X8 = false>
__global__ __launch_bounds__(256, 2) void convt2x2_kernel(ConvTArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int HW = a.H * a.W;
  // 1-D grid = N * pixel tiles (512 px) * virtual-channel groups (64 wide).  Workgroups b, b+8, ... share an XCD
  // (round-robin dispatch): give each XCD a contiguous run of work ids with the channel group fastest, so the
  // 4*Cout/64 workgroups that re-read one pixel tile find it in their own L2.  Speed only.
  const int nvg = (4 * a.Cout) >> 6;
  const int ptiles = (HW + 511) >> 9;
  const int G = (int)gridDim.x;
  const int slot = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  const int vg = slot % nvg;
  const int pt_n = slot / nvg;
  const int ptile = pt_n % ptiles;
  const int n = pt_n / ptiles;
  const int pbase = (ptile * 4 + wave) * 128;                 // first flattened pixel of this wave's tile
  if (pbase >= HW) return;                                    // whole wave out of range (no barriers in this kernel)
  const int vt0 = vg * 2;                                     // first 32-wide virtual-channel tile
  const int K16 = a.Cin >> 4;
  const int h = lane >> 5;

  float16v acc[4][2];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[pt][j][r] = 0.f;

  int pix[4];
#pragma unroll
  for (int pt = 0; pt < 4; ++pt) pix[pt] = min(pbase + pt * 32 + (lane & 31), HW - 1);   // clamp: discarded at the store
  const half_t* in_n = a.in + (size_t)n * a.Cin * HW * P;

  // operands of one K-step (16 input channels): 12 x 16-byte loads per lane in exact mode.  Two register sets:
  // the loads of step kk+1 are in flight under the 24 MFMAs of step kk.
  struct Ops { half8 wh[2], wl[2], xh[4], xl[4]; };
  auto load_ops = [&](Ops& o, int kk) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const half_t* w = a.wpk + (((size_t)(vt0 + j) * K16 + kk) * P) * 512 + lane * 8;
      o.wh[j] = *(const half8*)w;
      if (P == 2) o.wl[j] = *(const half8*)(w + 512);
    }
    const half_t* blk = in_n + (size_t)kk * HW * (P * 16) + h * 8;
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const half_t* x = blk + (size_t)pix[pt] * (P * 16);
      o.xh[pt] = *(const half8*)x;
      if (X8) {
        const u32x4 b8 = *(const u32x4*)(x + 16);                  // {lo8 c0-3, x8 c0-3, lo8 c4-7, x8 c4-7} of this lane's octet
        typedef _Float16 half2x __attribute__((ext_vector_type(2)));
        const half2x q0 = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(b8[0], X8_LO_MUL, false), q1 = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(b8[0], X8_LO_MUL, true);
        const half2x q2 = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(b8[2], X8_LO_MUL, false), q3 = __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(b8[2], X8_LO_MUL, true);
        o.xl[pt] = (half8){q0[0], q0[1], q1[0], q1[1], q2[0], q2[1], q3[0], q3[1]};
      } else if (P == 2) o.xl[pt] = *(const half8*)(x + 16);
    }
  };
  auto run_mfma = [&](const Ops& o) {
#pragma unroll
    for (int pt = 0; pt < 4; ++pt)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (P == 2) {
          acc[pt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh[j], o.xl[pt], acc[pt][j], 0, 0, 0);
          acc[pt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wl[j], o.xh[pt], acc[pt][j], 0, 0, 0);
        }
        acc[pt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(o.wh[j], o.xh[pt], acc[pt][j], 0, 0, 0);
      }
  };
  Ops o0, o1;
  load_ops(o0, 0);
  for (int kk = 0; kk < K16; kk += 2) {
    if (kk + 1 < K16) load_ops(o1, kk + 1);
    __builtin_amdgcn_sched_barrier(0);
    run_mfma(o0);
    __builtin_amdgcn_sched_barrier(0);
    if (kk + 1 < K16) {
      if (kk + 2 < K16) load_ops(o0, kk + 2);
      __builtin_amdgcn_sched_barrier(0);
      run_mfma(o1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: scale, bias (no activation), pixel shuffle, fp16 hi/lo packing, 16-byte stores
  const int H2 = 2 * a.H, W2 = 2 * a.W;
  const size_t oblk = (size_t)H2 * W2 * P * 16;                // halves per channel block of the output
  const int nbo = a.Cout >> 4;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int vb = (vt0 + j) * 32;                             // a 32-wide tile never straddles a (dy,dx) group: 32 | Cout
    const int q = vb / a.Cout, cbase = vb - q * a.Cout;
    const int dy = q >> 1, dx = q & 1;
    float sc[16], bi[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * h;
      sc[r] = a.scale[co]; bi[r] = a.bias[co];
    }
#pragma unroll
    for (int pt = 0; pt < 4; ++pt) {
      const int p = pbase + pt * 32 + (lane & 31);
      const bool ok = p < HW;
      const int pc = min(p, HW - 1);
      const int y = pc / a.W, x = pc - y * a.W;
      float v[16];
      float vmax = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = acc[pt][j][r] * sc[r] + bi[r];
#pragma unroll
      for (int r = 0; r < 16; r += 2) vmax = fmaxf(fmaxf(vmax, fabsf(v[r])), fabsf(v[r + 1]));
      if (__builtin_amdgcn_ballot_w64(vmax > F16_MAX)) {            // rare: report, then clamp to what fp16 can hold
        range_flag(a.status, vmax > F16_MAX, false);
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], -F16_MAX, F16_MAX);
      }
      half_t* dst = a.out + ((size_t)n * nbo + (cbase >> 4)) * oblk + ((size_t)(2 * y + dy) * W2 + (2 * x + dx)) * (P * 16);
      pack_store_octets<P, X8>(v, dst, oblk, ok, h);
    }
  }
}

// per-output-channel weight scale for a ConvTranspose2d weight [Cin][Cout][2][2]
__global__ void convt_scale_kernel(const float* __restrict__ w, int Cin, int Cout, const float* __restrict__ bias,
                                   float* __restrict__ mult_out, float* __restrict__ scale_out, unsigned* __restrict__ status) {
  const int co = blockIdx.x;
  float m = 0.f;
  bool bad = threadIdx.x == 0 && !(fabsf(bias[co]) <= 3.0e38f);
  for (int i = threadIdx.x; i < Cin * 4; i += blockDim.x) {
    const float x = fabsf(w[((size_t)(i >> 2) * Cout + co) * 4 + (i & 3)]);
    bad |= !(x <= 3.0e38f);
    m = fmaxf(m, x);
  }
  if (bad) atomicOr(status, ST_NAN);          // non-finite weights: the reference would carry NaN to its logits
  __shared__ float red[256];
  red[threadIdx.x] = m;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int e = 0, k = 0;
    const float mm = red[0];
    if (mm > 0.f && mm < 3.0e38f) { frexpf(mm, &e); k = 14 - e; }
    mult_out[co] = ldexpf(1.0f, k);
    scale_out[co] = ldexpf(1.0f, -k);
  }
}

// [Cin][Cout][2][2] fp32 -> [v-tile][Cin/16][P][64 lanes][8] fp16: lane l of the A fragment holds virtual channel
// vt*32 + (l & 31), input channels kk*16 + 8*(l >> 5) + 0..7
__global__ void convt_pack_kernel(const float* __restrict__ w, const float* __restrict__ mult, int Cin, int Cout, int P,
                                  half_t* __restrict__ out, long long units) {
  long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  long long t = u;
  const int lane = t % 64; t /= 64;
  const int pl = t % P; t /= P;
  const int K16 = Cin / 16;
  const int kk = t % K16; t /= K16;
  const int vt = (int)t;
  const int v = vt * 32 + (lane & 31);
  const int q = v / Cout, co = v - q * Cout;
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int ci = kk * 16 + 8 * (lane >> 5) + e;
    const float val = w[((size_t)ci * Cout + co) * 4 + q] * mult[co];
    const half_t hi = (half_t)val;
    r[e] = pl == 0 ? hi : (half_t)(val - (float)hi);
  }
  *(half8*)(out + u * 8) = r;
}

}  // namespace unetpp
