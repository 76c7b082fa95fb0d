// tapmm_ws.h — the decoder's first conv with FEWER FLOPS: conv3x3(cat[skip, up(low)]) split by linearity
// (reference src/models/unetpp.py:76,111-116: ConvBlock(cat([x_skip, self.up(x_low)]))):
//
//   conv(cat[skip, up(low)])[p] = conv_skip(skip)[p] + sum_tap W_tap . up(low)[p + tap]
//                               = conv_skip(skip)[p] + sum_tap up(W_tap . low)[p + tap]          (up is linear, per channel)
//
// i.e. the 2*Cout of the 3*Cout input channels that are bilinear x2 upsampled — two thirds of the layer's flops — are
// multiplied at LOW resolution (a quarter of the pixels), then interpolated:
//
//   1. tapmm_ws_kernel    Y[q][tap*Cout + co] = sum_c W_tap[co][c] * low[q][c]        1x1-conv GEMM, M = N*h*w low-res
//                         pixels, K = Cup, 9*Cout "virtual channels", raw fp32 accumulators (weights pre-scaled by the
//                         layer's per-channel power of two, hi/lo split operands, 3 MFMAs per product as everywhere)
//   2. upsum_kernel       Z[p][co] = sum_{tap: p+tap inside} bilinear(Y[.][tap*Cout + co])(p + tap)   fp32, align_corners=True
//   3. conv3x3_bias_relu_kernel<..., ZINIT>  over the skip channels only, accumulators initialised with Z
//
// Flops: 9*Cout*(Cs + Cup) per pixel -> 9*Cout*Cs + 9*Cout*Cup/4: half for Cup = 2 Cs (UNet++ decoder), -26 % of the
// whole network.  The zero padding is that of the reference: up(low) is zero outside the high-res image, so a tap that
// leaves the image contributes nothing.
//
// tapmm_ws_kernel is wave-specialised like conv3x3_ws.h: waves 0-3 multiply (128 pixels x 64 virtual channels each:
// 4 x 2 MFMA tiles, 2 x 2 waves = 256 x 128 per workgroup), waves 4-7 feed a ring of six 16-channel K-stages (24 KB
// each) by LDS-DMA, up to four stages in flight behind the one being multiplied; one LDS-only barrier per stage.
#pragma once
#include "conv3x3_ws.h"

namespace unetpp {

struct TapmmArgs {
  const half_t* low;     // [N][K/16][h][w][P][16]
  const half_t* wpk;     // [Nv/128][K/16][4 v-tiles][P][2][32][8]  (tapw_pack_kernel)
  float* y;              // [N][Nv/32][h*w][32] fp32 raw accumulators
  int N, hw, K, Nv;      // low-res pixels per image, input channels, virtual channels (9 * Cout)
#ifdef UNETPP_WS_DBG
  int dbg;               // measurement builds: 1 no DMA, 2 no MFMA, 4 no Y stores
#endif
};

struct TapmmCfg {
  static constexpr int NT = 512, NCONS = 4, NPROD = 4;
  static constexpr int TM = 256, TN = 128, MT = 4, NTL = 2;       // workgroup tile; MFMA tiles per consumer wave
  static constexpr int A_BYTES = TM * 64, B_BYTES = TN * 64, STAGE_BYTES = A_BYTES + B_BYTES;   // one 16-channel K-stage
  static constexpr int SLOTS = 6, LDS_BYTES = SLOTS * STAGE_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024, B_PIECES = B_BYTES / 1024;
  static constexpr int A_IT = A_PIECES / NPROD, B_IT = B_PIECES / NPROD, DMA_PER_STAGE = A_IT + B_IT;
  static constexpr int AHEAD = 4;                                   // stages in flight behind the published one
  static_assert(LDS_BYTES <= 160 * 1024 && AHEAD + 2 <= SLOTS, "ring");
};

// X8 (precision EXACT8, conv3x3_ws.h): the stages hold the 8-bit planes where the lo planes were -- pixels [lo8 | x8]
// (conv3x3_mfma.h), weights [wh8 | wl8] (tapw_pack_kernel) -- and the two cross terms of TWO consecutive stages (2 x 16
// channels x 2 kinds = K 64) go through one v_mfma_scale_f32_32x32x64_f8f6f4: bytes 0-15 of a lane from stage g, bytes 16-31
// from stage g + 1.  Per stage pair and output tile: 2 fp16 MFMAs + 1 scaled one (128 pipe cycles) instead of 6 (192).
template <bool X8>
__global__ __launch_bounds__(512, 2) void tapmm_ws_kernel(TapmmArgs a) {
  using C = TapmmCfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  touch_kernarg_lines<TapmmArgs>();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(unsigned long)(lds_char_t*)smem;
  const int mtiles_img = (a.hw + C::TM - 1) / C::TM;
  const int ntiles = a.Nv / C::TN;
  const int total_tiles = a.N * mtiles_img * ntiles;
  const int nst = a.K / 16;                                         // K-stages per tile
  const int G = (int)gridDim.x;
  const int slot = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  // tile t -> (image n, pixel tile, virtual-channel tile); the channel tile runs fastest so that the workgroups of
  // one XCD (consecutive slots) re-read the same 256 pixels from their own L2
  auto decode = [&](int t, int& n, int& q0, int& vt) {
    vt = t % ntiles; t /= ntiles;
    const int mt = t % mtiles_img;
    n = t / mtiles_img;
    q0 = mt * C::TM;
  };
  if (slot >= total_tiles) return;
  if (wave >= C::NCONS) {
    // =============================================================== producers
    const int pw = wave - C::NCONS;
    constexpr unsigned OOB = 0x80000000u;
    const unsigned blk_bytes = (unsigned)a.hw * 64u;                // one 16-channel block of one image
    // issue cursor: runs AHEAD + 1 stages in front of the publish cursor, across tile boundaries
    int it_tile = slot, it_s = 0;
    int in_, iq0, ivt;
    decode(it_tile, in_, iq0, ivt);
    auto issue = [&](int ring) {
#ifdef UNETPP_WS_DBG
      if (a.dbg & 1) { if (++it_s == nst) { it_s = 0; it_tile += G; if (it_tile < total_tiles) decode(it_tile, in_, iq0, ivt); } return; }
#endif
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.low + (size_t)in_ * (a.K / 16) * a.hw * 32), 0, (int)((a.K / 16) * blk_bytes), 0x00020000);
      const unsigned dstA = lds_base + ring * C::STAGE_BYTES;
#pragma unroll
      for (int it = 0; it < C::A_IT; ++it) {
        const int piece = pw + it * C::NPROD;                       // 16 pixels x 4 units, [unit][pixel][8 halves]
        const int q = iq0 + piece * 16 + (lane & 15);
        const unsigned voff = q < a.hw ? (unsigned)q * 64u + (unsigned)(lane >> 4) * 16u : OOB;
        blds16(rs, voff, it_s * (int)blk_bytes, dstA + piece * 1024);
      }
      const char* wsrc = (const char*)a.wpk + ((size_t)ivt * nst + it_s) * C::B_BYTES;
#pragma unroll
      for (int it = 0; it < C::B_IT; ++it) {
        const int piece = pw + it * C::NPROD;
        glds16(wsrc + piece * 1024, lane * 16, dstA + C::A_BYTES + piece * 1024);
      }
      if (++it_s == nst) {                                          // next tile of this workgroup
        it_s = 0; it_tile += G;
        if (it_tile < total_tiles) decode(it_tile, in_, iq0, ivt);
      }
    };
    int issued = 0, g = 0;                                          // stages issued / published so far
    for (int k = 0; k <= C::AHEAD && it_tile < total_tiles; ++k, ++issued) issue(issued % C::SLOTS);
    for (int tile = slot; tile < total_tiles; tile += G)
      for (int s = 0; s < nst; ++s, ++g) {
        // stages g and g+1 must have landed (the consumers read g and prefetch g+1's fragments); later ones may fly
        const int later = issued - (g + 2);
        if (later >= 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (later == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lds_barrier();                                              // publishes g, g+1; slot of stage g-1 is free
        if (it_tile < total_tiles) { issue(issued % C::SLOTS); ++issued; }
      }
    return;
  }

  // ================================================================= consumers
  const int cw = wave, pxh = cw & 1, vh = cw >> 1;
  static_assert(C::DMA_PER_STAGE == 6, "vmcnt ladder above");
  // fragment offsets inside a stage.  Pixel tile mt of this wave: pixels pxh*128 + mt*32 + (lane & 31)
  int a_off[C::MT];
#pragma unroll
  for (int mt = 0; mt < C::MT; ++mt) {
    const int px = pxh * 128 + mt * 32 + (lane & 31);
    a_off[mt] = (px >> 4) * 1024 + ((lane >> 5) * 16 + (px & 15)) * 16;          // unit = plane*2 + k-group; plane 1 = +512
  }
  const int b_off = C::A_BYTES + (vh * 2) * 2048 + (lane >> 5) * 512 + (lane & 31) * 16;   // v-tile vt: + vt*2048, plane 1: +1024
  struct Frag { half8 ah[C::MT], al[C::MT], bh[C::NTL], bl[C::NTL]; };
  auto load_frags = [&](Frag& f, int ring) {
    const char* st = smem + ring * C::STAGE_BYTES;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) { f.ah[mt] = *(const half8*)(st + a_off[mt]); f.al[mt] = *(const half8*)(st + a_off[mt] + 512); }
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt) { f.bh[nt] = *(const half8*)(st + b_off + nt * 2048); f.bl[nt] = *(const half8*)(st + b_off + nt * 2048 + 1024); }
  };
  float16v acc[C::MT][C::NTL];
  auto run_mfma = [&](const Frag& f) {
#ifdef UNETPP_WS_DBG
    if (a.dbg & 2) return;
#endif
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[nt], f.al[mt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[nt], f.ah[mt], acc[mt][nt], 0, 0, 0);
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[nt], f.ah[mt], acc[mt][nt], 0, 0, 0);
      }
  };

  int g = 0;
  Frag f0, f1;
  bool have_f0 = false;                                             // fragments of stage g already in f0 / f1 (parity of g)
  // X8: the fp16 hi fragments of a stage (h0: even stages, h1: odd ones) and the 8-bit operands of a stage PAIR
  struct HFrag { half8 a[C::MT], b[C::NTL]; };
  HFrag h0, h1;
  int8v xa[C::MT], xb[C::NTL];      // 8-bit operands of a stage pair: bytes 0-15 from stage g, bytes 16-31 from stage g + 1
  auto load_h = [&](HFrag& f, int ring) {
    const char* st = smem + ring * C::STAGE_BYTES;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) f.a[mt] = *(const half8*)(st + a_off[mt]);
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt) f.b[nt] = *(const half8*)(st + b_off + nt * 2048);
  };
  auto load_x = [&](int ring0, int ring1) {
    const char* s0 = smem + ring0 * C::STAGE_BYTES;
    const char* s1 = smem + ring1 * C::STAGE_BYTES;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
      const int4v lo = *(const int4v*)(s0 + a_off[mt] + 512), hi = *(const int4v*)(s1 + a_off[mt] + 512);
      xa[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
#pragma unroll
    for (int nt = 0; nt < C::NTL; ++nt) {
      const int4v lo = *(const int4v*)(s0 + b_off + nt * 2048 + 1024), hi = *(const int4v*)(s1 + b_off + nt * 2048 + 1024);
      xb[nt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto main_mfma = [&](const HFrag& f) {
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.b[nt], f.a[mt], acc[mt][nt], 0, 0, 0);
  };
  auto cross_mfma = [&]() {
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) {
        acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xb[nt], xa[mt], acc[mt][nt], 0 /* e4m3 */, 1 /* e5m2 */, 0, X8_SCALE_W, 0, X8_SCALE_A);
      }
  };
  for (int tile = slot; tile < total_tiles; tile += G) {
    int n, q0, vt;
    decode(tile, n, q0, vt);
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
    const bool last_tile = tile + G >= total_tiles;
    if constexpr (X8) {
      for (int s = 0; s < nst; s += 2) {
        lds_barrier();                                              // stages g, g+1 landed
        if (!have_f0) { load_h(h0, g % C::SLOTS); have_f0 = true; }
        load_x(g % C::SLOTS, (g + 1) % C::SLOTS);
        load_h(h1, (g + 1) % C::SLOTS);
        __builtin_amdgcn_sched_barrier(0);
        main_mfma(h0);
        __builtin_amdgcn_sched_barrier(0);
        ++g;
        lds_barrier();                                              // stages g, g+1 landed; every read of stage g-1 is done
        const bool more = !(last_tile && s + 2 >= nst);
        if (more) load_h(h0, (g + 1) % C::SLOTS);
        have_f0 = more;
        __builtin_amdgcn_sched_barrier(0);
        cross_mfma();                                               // the pair (g-1, g)
        main_mfma(h1);
        __builtin_amdgcn_sched_barrier(0);
        ++g;
      }
    } else
    for (int s = 0; s < nst; s += 2) {                              // two stages per trip: static register sets
      lds_barrier();                                                // stages g, g+1 landed
      if (!have_f0) { load_frags(f0, g % C::SLOTS); have_f0 = true; }
      load_frags(f1, (g + 1) % C::SLOTS);
      __builtin_amdgcn_sched_barrier(0);
      run_mfma(f0);
      __builtin_amdgcn_sched_barrier(0);
      ++g;
      lds_barrier();                                                // stages g, g+1 landed
      const bool more = !(last_tile && s + 2 >= nst);
      if (more) load_frags(f0, (g + 1) % C::SLOTS);
      have_f0 = more;
      __builtin_amdgcn_sched_barrier(0);
      run_mfma(f1);
      __builtin_amdgcn_sched_barrier(0);
      ++g;
    }
#ifdef UNETPP_WS_DBG
    if (a.dbg & 4) { if (acc[0][0][0] == 12345.f) a.y[0] = 1.f; continue; }
#endif
    // ---- store the raw accumulators: Y[n][v/32][q][32], lane = pixel, registers 4q'..4q'+3 = channels 8q' + 4h + (0..3)
    const int h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < C::MT; ++mt) {
      const int q = q0 + pxh * 128 + mt * 32 + (lane & 31);
#pragma unroll
      for (int nt = 0; nt < C::NTL; ++nt) {
        const int vb = vt * (C::TN / 32) + vh * 2 + nt;             // 32-wide virtual-channel block
        float* dst = a.y + (((size_t)n * (a.Nv / 32) + vb) * a.hw + q) * 32 + 4 * h;
        if (q < a.hw) {
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            typedef __attribute__((ext_vector_type(4))) float f32x4;
            *(f32x4*)(dst + 8 * qq) = (f32x4){acc[mt][nt][4 * qq], acc[mt][nt][4 * qq + 1], acc[mt][nt][4 * qq + 2], acc[mt][nt][4 * qq + 3]};
          }
        }
      }
    }
  }
}

// canonical conv1 weight [Cout][Cs + Cup][3][3] fp32 -> tapmm's A operand (the up channels only):
// [v-tile 128][k16][vt 4][plane][kg 2][32 v][8 halves], virtual channel v = tap * Cout + co, scaled by mult[co]
// x8: plane 1 holds, per (h = former k-group, v), the 16 e4m3 bytes [wh8 x 4 | wl8 x 4 | wh8 x 4 | wl8 x 4] of input channels
// 16 ks + 8 h + 0..7 (weight_pack_x8_kernel's encoding: wh8 = e4m3(2^-6 ws), wl8 = e4m3(2^5 (ws - fp16(ws))))
__global__ void tapw_pack_kernel(const float* __restrict__ w, const float* __restrict__ mult, int Cout, int Cs, int Cup,
                                 half_t* __restrict__ out, long long units, int x8) {
  const long long u = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (u >= units) return;
  long long t = u;
  const int vv = t % 32; t /= 32;
  const int kg = t % 2; t /= 2;
  const int pl = t % 2; t /= 2;
  const int vt = t % 4; t /= 4;
  const int nst = Cup / 16;
  const int ks = t % nst; t /= nst;
  const int v128 = (int)t;
  const int v = v128 * 128 + vt * 32 + vv;
  const int tap = v / Cout, co = v - tap * Cout;
  half8 r;
  float val[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = ks * 16 + kg * 8 + e;
    val[e] = w[((size_t)co * (Cs + Cup) + Cs + c) * 9 + tap] * mult[co];
    const half_t hi = (half_t)val[e];
    r[e] = pl == 0 ? hi : (half_t)(val[e] - (float)hi);
  }
  if (x8 && pl == 1) {
    unsigned wd[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int a8 = 0, b8 = 0;
      float hi4[4], lo4[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { hi4[i] = val[4 * q + i] * 0.015625f; lo4[i] = (val[4 * q + i] - (float)(half_t)val[4 * q + i]) * 32.0f; }
      a8 = __builtin_amdgcn_cvt_pk_fp8_f32(hi4[0], hi4[1], a8, false); a8 = __builtin_amdgcn_cvt_pk_fp8_f32(hi4[2], hi4[3], a8, true);
      b8 = __builtin_amdgcn_cvt_pk_fp8_f32(lo4[0], lo4[1], b8, false); b8 = __builtin_amdgcn_cvt_pk_fp8_f32(lo4[2], lo4[3], b8, true);
      wd[2 * q] = (unsigned)a8; wd[2 * q + 1] = (unsigned)b8;
    }
    *(u32x4*)(out + u * 8) = (u32x4){wd[0], wd[1], wd[2], wd[3]};
    return;
  }
  *(half8*)(out + u * 8) = r;
}

// Z[n][co/32][p][32] = sum over the taps whose pixel p + tap lies inside the H x W image of the bilinear x2
// (align_corners=True, src = dst * (in-1)/(out-1), weights as upsample2x_kernel) interpolation of Y's channel
// tap*Cout + co at p + tap; fp32 throughout.
struct UpsumArgs {
  const float* y;        // [N][9*Cout/32][h*w][32]
  float* z;              // [N][Cout/32][H*W][32]
  int N, H, W, Cout;
};

// Tile = 16 x 32 output pixels x 32 channels per workgroup (256 threads: thread = one column x 4 channels, 16 rows).
// Per tap the <= 11 x 19 low-res records (128 B each) the tile can touch arrive by LDS-DMA, so every Y record is fetched
// once per tile and tap and the 36 corner reads per output come from LDS, not from the vector cache.  One staging
// buffer and <= 128 VGPRs: four workgroups share a CU and cover each other's DMA round trips.
template <int TH_>
struct UpsumCfgT {
  static constexpr int TH = TH_, TW = 32, RH = TH / 2 + 3, RW = TW / 2 + 3, RECS = RH * RW;
  static constexpr int PIECES = (RECS * 128 + 1023) / 1024, BUF_BYTES = PIECES * 1024;
  static constexpr int YTAB_BYTES = (TH + 2) * 16, LDS_BYTES = BUF_BYTES + YTAB_BYTES;     // one buffer: four workgroups per CU overlap each other
  static constexpr int ITERS = (PIECES + 3) / 4;
};
typedef UpsumCfgT<16> UpsumCfg;

// TH = rows of a tile: 16, or 4 for grids that would leave most CUs idle (small batches: a workgroup's 16 rows x 9 taps of corner
// reads and FMAs are ~20 us of dependent work however few workgroups there are).  TB = taps staged per round (1; 3 was tried for
// the small grids and changed nothing: the launch waits for the workgroups' own arithmetic, not for their DMA round trips).
template <int TB, int TH_ = 16>
__global__ __launch_bounds__(256, TB == 1 ? 4 : 1) void upsum_kernel(UpsumArgs a) {
  using C = UpsumCfgT<TH_>;
  typedef __attribute__((ext_vector_type(4))) float f32x4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = a.H, W = a.W, h = H >> 1, w = W >> 1;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds_base = (unsigned)(unsigned long)(lds_char_t*)smem;
  const int tiles_x = (W + C::TW - 1) / C::TW, tiles_y = (H + C::TH - 1) / C::TH;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; t /= tiles_y;
  const int cb = t % (a.Cout / 32);
  const int n = t / (a.Cout / 32);
  const int y0 = ty * C::TH, x0 = tx * C::TW;
  const float sh = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  // low-res origin of the staged region: first row / column any pixel y0-1 .. y0+TH, x0-1 .. x0+TW touches
  const int ybase = (int)(sh * (float)max(y0 - 1, 0)), xbase = (int)(sw * (float)max(x0 - 1, 0));
  const int vblocks = 9 * a.Cout / 32;
  const size_t plane = (size_t)h * w * 128;                        // bytes of one (n, 32-channel block) plane of Y
  // this lane's records in its DMA pieces: record = 8 per piece, 16-byte part = lane & 7
  unsigned voff[C::ITERS];
#pragma unroll
  for (int it = 0; it < C::ITERS; ++it) {
    const int rec = (wave + it * 4) * 8 + (lane >> 3);
    const int ry = rec / C::RW, rx = rec - ry * C::RW;
    const int yy = min(ybase + ry, h - 1), xx = min(xbase + rx, w - 1);       // clamped: always a valid record
    voff[it] = (wave + it * 4 < C::PIECES) ? (unsigned)((yy * w + xx) * 128 + (lane & 7) * 16) : 0x80000000u;
  }
  auto stage = [&](int tap, int buf) {
    const int vb = (tap * a.Cout) / 32 + cb;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)a.y + ((size_t)n * vblocks + vb) * plane), 0,
                                                                       (int)plane, 0x00020000);
#pragma unroll
    for (int it = 0; it < C::ITERS; ++it) {
      const int piece = wave + it * 4;
      if (piece < C::PIECES) blds16(rs, voff[it], 0, lds_base + buf * C::BUF_BYTES + piece * 1024);
    }
  };
  // thread = column x0 + (tid >> 3), channel quad tid & 7; per shifted column (dx) and per shifted row: corner offsets + weights
  const int cq = tid & 7, col = tid >> 3;
  int xo0[3], xo1[3]; float xl0[3], xl1[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int xx = x0 + col + dx - 1;
    const bool in = xx >= 0 && xx < W && x0 + col < W;
    const float fx = sw * (float)max(xx, 0);
    const int c0 = min((int)fx, w - 1), c1 = c0 + (c0 < w - 1 ? 1 : 0);
    const float l1 = fminf(fmaxf(fx - (float)c0, 0.f), 1.f);
    xo0[dx] = min(max(c0 - xbase, 0), C::RW - 1) * 128 + cq * 16;
    xo1[dx] = min(max(c1 - xbase, 0), C::RW - 1) * 128 + cq * 16;
    xl0[dx] = in ? 1.f - l1 : 0.f; xl1[dx] = in ? l1 : 0.f;      // a tap column outside the image contributes nothing
  }
  // per shifted row (the same for every thread): corner row offsets + weights, kept in LDS
  struct YRow { int o0, o1; float l0, l1; };
  YRow* ytab = (YRow*)(smem + TB * C::BUF_BYTES);
  if (tid < C::TH + 2) {
    const int yy = y0 + tid - 1;
    const bool in = yy >= 0 && yy < H;
    const float fy = sh * (float)max(yy, 0);
    const int r0 = min((int)fy, h - 1), r1 = r0 + (r0 < h - 1 ? 1 : 0);
    const float l1 = fminf(fmaxf(fy - (float)r0, 0.f), 1.f);
    YRow e;
    e.o0 = min(max(r0 - ybase, 0), C::RH - 1) * (C::RW * 128);
    e.o1 = min(max(r1 - ybase, 0), C::RH - 1) * (C::RW * 128);
    e.l0 = in ? 1.f - l1 : 0.f; e.l1 = in ? l1 : 0.f;
    ytab[tid] = e;
  }
  f32x4 sum[C::TH];
#pragma unroll
  for (int r = 0; r < C::TH; ++r) sum[r] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int tap0 = 0; tap0 < 9; tap0 += TB) {
    __syncthreads();                                                // everyone has left the buffers (first trip: ytab visible)
#pragma unroll
    for (int t = 0; t < TB; ++t) stage(tap0 + t, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                                // the taps' records landed
#pragma unroll
    for (int t = 0; t < TB; ++t) {
    const int tap = tap0 + t;
    const char* reg = smem + t * C::BUF_BYTES;
    const int dy = tap / 3, dx = tap - dy * 3;
#pragma unroll
    for (int r = 0; r < C::TH; ++r) {
      const YRow yr = ytab[r + dy];                                 // image row y0 + r + dy - 1
      const f32x4 c00 = *(const f32x4*)(reg + yr.o0 + xo0[dx]), c01 = *(const f32x4*)(reg + yr.o0 + xo1[dx]);
      const f32x4 c10 = *(const f32x4*)(reg + yr.o1 + xo0[dx]), c11 = *(const f32x4*)(reg + yr.o1 + xo1[dx]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t0 = fmaf(xl1[dx], c01[e], xl0[dx] * c00[e]), t1 = fmaf(xl1[dx], c11[e], xl0[dx] * c10[e]);   // x inside each row first
        sum[r][e] += fmaf(yr.l1, t1, yr.l0 * t0);
      }
    }
    }
  }
  const int x = x0 + col;
  if (x < W) {
#pragma unroll
    for (int r = 0; r < C::TH; ++r)
      if (y0 + r < H)
        *(f32x4*)(a.z + (((size_t)n * (a.Cout / 32) + cb) * ((size_t)H * W) + (size_t)(y0 + r) * W + x) * 32 + cq * 4) = sum[r];
  }
}

}  // namespace unetpp
