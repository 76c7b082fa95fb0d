// conv3x3_mfma.h — 3x3 / pad 1 / stride 1 convolution + folded-BN bias + ReLU as an LDS-tiled
// implicit GEMM on gfx950 MFMA (v_mfma_f32_32x32x16_f16), NHWC activations.
//
// Replaces ConvBlock's  relu(bn(conv(x)))  (reference src/models/unetpp.py:23-26) for one conv, and,
// through its two-source loader, the torch.cat([skip, up], 1) in front of the decoder blocks
// (unetpp.py:111-116): channels [0,C0) come from in0, [C0,C0+C1) from in1 — the concatenated tensor is
// never materialised.
//
// GEMM view:  D[pixel][cout] = sum_{tap,cin} A[pixel+tap][cin] * W[tap][cin][cout]
//   M = pixels of a TH x 32 spatial tile (one MFMA row-tile = 32 consecutive x of one image row),
//       TH = WAVES*MW rows, every wave owns MW rows x all BN channels
//   N = BN = 32*NW output channels,  K = 9 taps x Cin, walked in chunks of KC input channels.
// Per chunk the workgroup holds in LDS (double buffered, one barrier per chunk):
//   * the input halo  (TH+2) x 34 pixels x KC channels — loaded to registers while the previous chunk
//     computes (bounds / zero padding handled there), written to the other buffer afterwards;
//   * the weight slab 9 x KC x BN — pre-packed in exactly the order the B fragments are read, so it is
//     a linear copy done by LDS-DMA (global_load_lds_dwordx4), in flight during the MFMAs.
// Activation layout in HBM: [N][H][W][P][C] fp16, P = 1 (FAST) or 2 (EXACT: plane 0 = hi, plane 1 = lo,
// value = hi + lo).  EXACT issues three MFMAs per product (lo*hi, hi*lo, hi*hi) into one fp32 accumulator.
//
// LDS images (bytes):
//   halo   [P][KG=KC/8][halo pixel][8 halves]  plane stride KGS == 32 (mod 128): the 8-lane groups of
//          ds_write_b128 (2 pixels x 4 channel groups) hit distinct banks; the A-fragment ds_read_b128 of
//          a 32x16 tile reads 2 x 512 contiguous bytes -> conflict-free.
//   slab   [P][tap][KG][BN][8 halves]           B-fragment read = 2 x 512 contiguous bytes.
//   The epilogue tile [pixel][P][EPN] fp16 reuses the staging memory after the last chunk: global stores
//   are full 16-byte lanes on contiguous channel runs and the 2x2 max-pool (unetpp.py:75) is taken
//   from it without another pass over HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unetpp {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float float16v;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ConvArgs {
  const half_t* in0;     // [N][H][W][P][C0]
  const half_t* in1;     // [N][H][W][P][C1] or nullptr (then C1 = 0)
  const half_t* wpk;     // packed weights [ct][chunk][P][tap][KG][BN][8]
  const float* scale;    // [Cout] 2^-k undoing the per-channel weight scaling
  const float* bias;     // [Cout] folded conv+BN bias
  half_t* out;           // [N][H][W][P][Cout]
  half_t* pool_out;      // [N][H/2][W/2][P][Cout] or nullptr
  int N, H, W, C0, C1, Cout;
  int tiles_x, tiles_y;  // spatial tiles per image
  int nct;               // Cout / BN
  int nchunks;           // ceil(C0 / KC) + C1 / KC
};

__host__ __device__ constexpr int conv_kgs(int nhalo) {
  // smallest value >= nhalo*16 that is == 32 (mod 128)
  int b = nhalo * 16;
  int r = ((32 - (b % 128)) + 128) % 128;
  return b + r;
}

template <int P, int KC, int NW, int MW, int WAVES>
struct ConvCfg {
  static constexpr int NT = WAVES * 64;
  static constexpr int TH = WAVES * MW, TW = 32, HALO_W = TW + 2, NHALO = (TH + 2) * HALO_W;
  static constexpr int KG = KC / 8, BN = 32 * NW;
  static constexpr int KGS = conv_kgs(NHALO);
  static constexpr int HALO_BYTES = P * KG * KGS;
  static constexpr int SLAB_BYTES = P * 9 * KC * BN * 2;
  static constexpr int BUF_BYTES = HALO_BYTES + SLAB_BYTES;
  static constexpr int EPN = BN < 64 ? BN : 64;
  static constexpr int EP_BYTES = TH * TW * P * EPN * 2;
  static constexpr int STAGE_BYTES = 2 * BUF_BYTES;
  static constexpr int LDS_BYTES = STAGE_BYTES > EP_BYTES ? STAGE_BYTES : EP_BYTES;
  static constexpr int HALO_ITEMS = NHALO * P * KG;
  static constexpr int HALO_ITERS = (HALO_ITEMS + NT - 1) / NT;
  static constexpr int SLAB_PIECES = SLAB_BYTES / 1024;   // one LDS-DMA wave-instruction = 1 KiB
  static_assert(SLAB_BYTES % 1024 == 0, "slab must be a whole number of 1 KiB DMA pieces");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ void split_f16(float v, half_t& hi, half_t& lo) {
  v = fminf(v, 65504.0f);
  hi = (half_t)v;
  lo = (half_t)(v - (float)hi);
}

typedef __attribute__((address_space(3))) char lds_char_t;

// LDS-DMA of 64 lanes x 16 B: LDS destination = wave-uniform byte address `lds_dst` + lane*16, global
// source per lane.  Issued from inline asm on purpose: hipcc (ROCm 7.2) treats the builtin form as a
// flat access that may alias LDS and then degrades every later `s_waitcnt lgkmcnt(N)` of the MFMA loop
// to lgkmcnt(0) (no ds_read prefetch overlap).  The asm is invisible to the compiler's counters, so
// the consumer side waits explicitly: `s_waitcnt vmcnt(0)` before the barrier that publishes the buffer.
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

template <int P, int KC, int NW, int MW, int WAVES, bool POOL>
__global__ __launch_bounds__(WAVES * 64) void conv3x3_bias_relu_kernel(ConvArgs a) {
  using C = ConvCfg<P, KC, NW, MW, WAVES>;
  constexpr int NT = C::NT, TH = C::TH, TW = C::TW, HALO_W = C::HALO_W;
  constexpr int KG = C::KG, BN = C::BN, KGS = C::KGS, EPN = C::EPN;
  constexpr int ITERS = C::HALO_ITERS;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int bid = blockIdx.x;
  const int ct = bid % a.nct;
  int pt = bid / a.nct;
  const int tx = pt % a.tiles_x; pt /= a.tiles_x;
  const int ty = pt % a.tiles_y;
  const int n = pt / a.tiles_y;
  const int x0 = tx * TW, y0 = ty * TH;
  const int H = a.H, W = a.W;

  float16v acc[MW][NW];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

  // ---- chunk-invariant halo item geometry (per thread: ITERS items of 16 bytes).  The loads are raw
  // buffer loads: one descriptor per source covering image n, per-lane byte offset precomputed here,
  // the chunk's channel offset in the scalar soffset; pixels outside the image (zero padding) and
  // unused items carry an out-of-range offset and read back zeros — no branches in the K loop.
  constexpr unsigned OOB = 0x80000000u;
  unsigned voff0[ITERS], voff1[ITERS];
  int ldsoff[ITERS];   // byte offset inside the halo image, -1 = unused item
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int i = tid + it * NT;
    const int kg = i % KG;
    const int pl = (i / KG) % P;
    const int hp = i / (KG * P);
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    const int gy = y0 + hy - 1, gx = x0 + hx - 1;
    const bool ok = (i < C::HALO_ITEMS) && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const unsigned pix = (unsigned)((gy * W + gx) * P + pl);
    voff0[it] = (ok && kg * 8 < a.C0) ? (pix * a.C0 + kg * 8) * 2u : OOB;
    voff1[it] = ok ? (pix * a.C1 + kg * 8) * 2u : OOB;
    ldsoff[it] = (i < C::HALO_ITEMS) ? (pl * KG + kg) * KGS + hp * 16 : -1;
  }
  const size_t img = (size_t)n * H * W * P;
  const unsigned img_bytes0 = (unsigned)(H * W * P * a.C0 * 2), img_bytes1 = (unsigned)(H * W * P * a.C1 * 2);
  const __amdgpu_buffer_rsrc_t rsrc0 =
      __builtin_amdgcn_make_buffer_rsrc((void*)(a.in0 + img * a.C0), 0, (int)img_bytes0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrc1 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(a.in1 ? a.in1 + img * a.C1 : a.in0), 0, (int)(a.in1 ? img_bytes1 : 0u), 0x00020000);
  const int nch0 = (a.C0 + KC - 1) / KC;
  const char* wsrc = (const char*)a.wpk + (size_t)ct * a.nchunks * C::SLAB_BYTES;

  u32x4 hv[ITERS];
  auto halo_issue_one = [&](int c, int it) {     // `it` is a compile-time constant at every call site
    if (c < nch0) hv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc0, (int)voff0[it], c * KC * 2, 0);
    else hv[it] = __builtin_amdgcn_raw_buffer_load_b128(rsrc1, (int)voff1[it], (c - nch0) * KC * 2, 0);
  };
  auto halo_commit = [&](char* halo) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it)
      if (ldsoff[it] >= 0) *(u32x4*)(halo + ldsoff[it]) = hv[it];
  };
  const unsigned lds_base = (unsigned)(unsigned long)(lds_char_t*)smem;
  constexpr int DMA_PER_WAVE = (C::SLAB_PIECES + WAVES - 1) / WAVES;
  auto slab_dma_one = [&](int c, int slab_off, int p) {
    const int piece = wave + p * WAVES;
    if (C::SLAB_PIECES % WAVES == 0 || piece < C::SLAB_PIECES)
      glds16(wsrc + (size_t)c * C::SLAB_BYTES + piece * 1024, lane * 16, lds_base + slab_off + piece * 1024);
  };

  // lane-constant LDS read offsets
  const int a_lane_off = (lane >> 5) * KGS + ((wave * MW) * HALO_W + (lane & 31)) * 16;
  const int b_lane_off = ((lane >> 5) * BN + (lane & 31)) * 16;

  struct Frag { half8 ah[MW], al[MW], bh[NW], bl[NW]; };
  auto load_frags = [&](Frag& f, const char* halo, const char* slab, int step) {
    const int tap = step % 9, s = step / 9;
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      const int off = a_lane_off + (2 * s) * KGS + ((m + dy) * HALO_W + dx) * 16;
      f.ah[m] = *(const half8*)(halo + off);
      if (P == 2) f.al[m] = *(const half8*)(halo + off + KG * KGS);
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int off = b_lane_off + ((tap * KG + 2 * s) * BN + j * 32) * 16;
      f.bh[j] = *(const half8*)(slab + off);
      if (P == 2) f.bl[j] = *(const half8*)(slab + off + 9 * KC * BN * 2);
    }
  };
  auto run_mfma = [&](const Frag& f) {
#ifdef UNETPP_ABLATE_MFMA   // dev-only timing build: keep the fragments live, issue no MFMA
#pragma unroll
    for (int m = 0; m < MW; ++m) { asm volatile("" ::"v"(f.ah[m])); if (P == 2) asm volatile("" ::"v"(f.al[m])); }
#pragma unroll
    for (int j = 0; j < NW; ++j) { asm volatile("" ::"v"(f.bh[j])); if (P == 2) asm volatile("" ::"v"(f.bl[j])); }
    return;
#endif
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        if (P == 2) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[m], f.bh[j], acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bl[j], acc[m][j], 0, 0, 0);
        }
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[m], f.bh[j], acc[m][j], 0, 0, 0);
      }
  };

  // ---- prologue: chunk 0 -> buffer 0
#pragma unroll
  for (int it = 0; it < ITERS; ++it) halo_issue_one(0, it);
#pragma unroll
  for (int p = 0; p < DMA_PER_WAVE; ++p) slab_dma_one(0, C::HALO_BYTES, p);
  halo_commit(smem);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifdef UNETPP_STAMP
  unsigned long long st_issue = 0, st_mfma = 0, st_commit = 0, st_barrier = 0, st_t;
#define STAMP(acc_) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); acc_ += t_ - st_t; st_t = t_; } while (0)
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_t) :: "memory");
#else
#define STAMP(acc_)
#endif
  for (int c = 0; c < a.nchunks; ++c) {
    char* cur = smem + (c & 1) * C::BUF_BYTES;
    char* nxt = smem + ((c & 1) ^ 1) * C::BUF_BYTES;
#ifdef UNETPP_ABLATE_GLOBAL   // dev-only timing build: no global traffic after the prologue
    const bool more = false;
#else
    const bool more = c + 1 < a.nchunks;
#endif
    const int nxt_slab = ((c & 1) ^ 1) * C::BUF_BYTES + C::HALO_BYTES;
    // ---- MFMA over 9 taps x KC from the current buffer.  One step = one tap x 16 channels; the
    // fragments of step t+1 are read from LDS (into the other register set) before the MFMAs of
    // step t issue, so the LDS latency hides under the matrix pipe.  The next chunk's global loads
    // (one halo item + one DMA piece per step) are spread over the first steps instead of being
    // issued in one burst, and the halo registers are committed to the other buffer just before the
    // last step's MFMAs, so neither the vector-memory issue queue nor the LDS writes idle the pipe.
    const char* halo = cur;
    const char* slab = cur + C::HALO_BYTES;
    constexpr int NSTEPS = 9 * (KC / 16);
    constexpr int HPS = (ITERS + NSTEPS - 2) / (NSTEPS - 1);          // halo items issued per step
    constexpr int DPS = (DMA_PER_WAVE + NSTEPS - 2) / (NSTEPS - 1);   // DMA pieces issued per step
    Frag f0, f1;
    load_frags(f0, halo, slab, 0);
#pragma unroll
    for (int st = 0; st < NSTEPS; ++st) {
      Frag& fc = (st & 1) ? f1 : f0;
      Frag& fn = (st & 1) ? f0 : f1;
      if (st + 1 < NSTEPS) load_frags(fn, halo, slab, st + 1);
      if (more) {
#pragma unroll
        for (int k = st * HPS; k < (st + 1) * HPS; ++k)
          if (k < ITERS) halo_issue_one(c + 1, k);
#pragma unroll
        for (int k = st * DPS; k < (st + 1) * DPS; ++k)
          if (k < DMA_PER_WAVE) slab_dma_one(c + 1, nxt_slab, k);
        if (st == NSTEPS - 1) halo_commit(nxt);
      }
      __builtin_amdgcn_sched_barrier(0);      // keep prefetch + load issue ahead of this step's MFMAs
      run_mfma(fc);
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(st_mfma);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
    STAMP(st_commit);
    __syncthreads();
    STAMP(st_barrier);
  }

#ifdef UNETPP_STAMP
  if (blockIdx.x == 7 && lane == 0) {
    unsigned long long* dbg = (unsigned long long*)a.pool_out + wave * 4;
    dbg[0] = st_issue; dbg[1] = st_mfma; dbg[2] = st_commit; dbg[3] = st_barrier;
  }
#endif
  // ---- epilogue: scale, bias, ReLU -> LDS tile -> coalesced stores (+ fused 2x2 max-pool)
  half_t* ep = (half_t*)smem;
  const int Cout = a.Cout;
#pragma unroll
  for (int j0 = 0; j0 < NW; j0 += EPN / 32) {
    if (j0) __syncthreads();
#pragma unroll
    for (int jj = 0; jj < EPN / 32; ++jj) {
      const int j = j0 + jj;
      const int co = ct * BN + j * 32 + (lane & 31);
      const float sc = a.scale[co], bi = a.bias[co];
#pragma unroll
      for (int m = 0; m < MW; ++m) {
        const int row = wave * MW + m;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int x = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          float v = fmaxf(acc[m][j][r] * sc + bi, 0.f);
          const int idx = ((row * TW + x) * P) * EPN + jj * 32 + (lane & 31);
          if (P == 2) {
            half_t hi, lo;
            split_f16(v, hi, lo);
            ep[idx] = hi;
            ep[idx + EPN] = lo;
          } else {
            ep[idx] = (half_t)fminf(v, 65504.0f);
          }
        }
      }
    }
    __syncthreads();
    {
      constexpr int CU = EPN / 8;
      constexpr int UNITS = TH * TW * P * CU;
      static_assert(UNITS % NT == 0, "epilogue units");
      half_t* out_n = a.out + (size_t)n * H * W * P * Cout;
#pragma unroll
      for (int it = 0; it < UNITS / NT; ++it) {
        int u = tid + it * NT;
        int cu = u % CU;
        int pl = (u / CU) % P;
        int px = u / (CU * P);
        int gy = y0 + px / TW, gx = x0 + px % TW;
        if (gy < H && gx < W)
          *(u32x4*)(out_n + ((size_t)(gy * W + gx) * P + pl) * Cout + ct * BN + j0 * 32 + cu * 8) =
              *(const u32x4*)(ep + (size_t)u * 8);
      }
    }
    if (POOL) {
      constexpr int CU = EPN / 8;
      constexpr int PUNITS = (TH / 2) * (TW / 2) * CU;
      const int Hp = H >> 1, Wp = W >> 1;
      half_t* pool_n = a.pool_out + (size_t)n * Hp * Wp * P * Cout;
      for (int u = tid; u < PUNITS; u += NT) {
        int cu = u % CU;
        int pp = u / CU;
        int py = pp / (TW / 2), px = pp % (TW / 2);
        int gy = (y0 >> 1) + py, gx = (x0 >> 1) + px;
        if (gy >= Hp || gx >= Wp) continue;
        const half_t* p00 = ep + ((size_t)((2 * py) * TW + 2 * px) * P) * EPN + cu * 8;
        const int dxs = P * EPN, dys = TW * P * EPN;
        half_t* dst = pool_n + ((size_t)(gy * Wp + gx) * P) * Cout + ct * BN + j0 * 32 + cu * 8;
        if (P == 1) {
          half8 q0 = *(const half8*)p00, q1 = *(const half8*)(p00 + dxs);
          half8 q2 = *(const half8*)(p00 + dys), q3 = *(const half8*)(p00 + dys + dxs);
          half8 r;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            half_t m0 = q0[e] > q1[e] ? q0[e] : q1[e];
            half_t m1 = q2[e] > q3[e] ? q2[e] : q3[e];
            r[e] = m0 > m1 ? m0 : m1;
          }
          *(half8*)dst = r;
        } else {
          half8 rh, rl;
          half8 h0 = *(const half8*)p00, l0 = *(const half8*)(p00 + EPN);
          half8 h1 = *(const half8*)(p00 + dxs), l1 = *(const half8*)(p00 + dxs + EPN);
          half8 h2 = *(const half8*)(p00 + dys), l2 = *(const half8*)(p00 + dys + EPN);
          half8 h3 = *(const half8*)(p00 + dys + dxs), l3 = *(const half8*)(p00 + dys + dxs + EPN);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            float v0 = (float)h0[e] + (float)l0[e], v1 = (float)h1[e] + (float)l1[e];
            float v2 = (float)h2[e] + (float)l2[e], v3 = (float)h3[e] + (float)l3[e];
            float v = fmaxf(fmaxf(v0, v1), fmaxf(v2, v3));
            half_t hi, lo;
            split_f16(v, hi, lo);
            rh[e] = hi;
            rl[e] = lo;
          }
          *(half8*)dst = rh;
          *(half8*)(dst + Cout) = rl;
        }
      }
    }
  }
}

}  // namespace unetpp
