// conv3x3_mfma.h — 3x3 / pad 1 / stride 1 convolution + folded-BN bias + ReLU as an LDS-tiled
// implicit GEMM on gfx950 MFMA (v_mfma_f32_32x32x16_f16), NHWC activations.
//
// Replaces ConvBlock's  relu(bn(conv(x)))  (reference src/models/unetpp.py:23-26) for one conv.
//
// GEMM view:  D[pixel][cout] = sum_{tap,cin} A[pixel+tap][cin] * W[tap][cin][cout]
//   M = pixels of a TH x 32 spatial tile (one MFMA row-tile = 32 consecutive x of one image row)
//   N = BN = 32*NW output channels,  K = 9 taps x Cin, walked in chunks of KC input channels.
// Per chunk a workgroup (4 waves) stages in LDS
//   * the input halo  (TH+2) x 34 pixels x KC channels   — reused by all 9 taps,
//   * the weight slab 9 x KC x BN                        — pre-packed on the host side of the ABI in
//     exactly the order the B fragments are read, so the copy is linear and conflict-free.
// Activation layout in HBM: [N][H][W][P][C] fp16, P = 1 (FAST) or 2 (EXACT: plane 0 = hi, plane 1 = lo,
// value = hi + lo).  EXACT issues three MFMAs per product (lo*hi, hi*lo, hi*hi) into one fp32 accumulator.
//
// LDS images (bytes):
//   halo   [P][KG=KC/8][halo pixel][8 halves]  plane stride KGS == 32 (mod 128) so the 8-lane groups of
//          ds_write_b128 (2 pixels x 4 channel groups) hit distinct banks; A-fragment ds_read_b128 of a
//          32x16 tile reads 2 x 512 contiguous bytes -> conflict-free.
//   slab   [P][tap][KG][BN][8 halves]           B-fragment read = 2 x 512 contiguous bytes.
//   epilogue tile [pixel][P][EPN] fp16 reuses the same memory after the last chunk, so that global
//          stores are full 16-byte lanes on contiguous channel runs and the 2x2 max-pool
//          (reference unetpp.py:75) can be taken from it without a second pass over HBM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unetpp {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(16))) float float16v;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ConvArgs {
  const half_t* in;      // [N][H][W][P][Cin]
  const half_t* wpk;     // packed weights [ct][chunk][P][tap][KG][BN][8]
  const float* scale;    // [Cout] 2^-k undoing the per-channel weight scaling
  const float* bias;     // [Cout] folded conv+BN bias
  half_t* out;           // [N][H][W][P][Cout]
  half_t* pool_out;      // [N][H/2][W/2][P][Cout] or nullptr
  int N, H, W, Cin, Cout;
  int tiles_x, tiles_y;  // spatial tiles per image
  int nct;               // Cout / BN
  int nchunks;           // ceil(Cin / KC)
};

__host__ __device__ constexpr int conv_kgs(int nhalo) {
  // smallest value >= nhalo*16 that is == 32 (mod 128)
  int b = nhalo * 16;
  int r = ((32 - (b % 128)) + 128) % 128;
  return b + r;
}

template <int P, int KC, int NW, int MW>
struct ConvCfg {
  static constexpr int TH = 4 * MW, TW = 32, HALO_W = TW + 2, NHALO = (TH + 2) * HALO_W;
  static constexpr int KG = KC / 8, BN = 32 * NW;
  static constexpr int KGS = conv_kgs(NHALO);
  static constexpr int HALO_BYTES = P * KG * KGS;
  static constexpr int SLAB_BYTES = P * 9 * KC * BN * 2;
  static constexpr int EPN = BN < 64 ? BN : 64;
  static constexpr int EP_BYTES = TH * TW * P * EPN * 2;
  static constexpr int STAGE_BYTES = HALO_BYTES + SLAB_BYTES;
  static constexpr int LDS_BYTES = STAGE_BYTES > EP_BYTES ? STAGE_BYTES : EP_BYTES;
};

__device__ __forceinline__ void split_f16(float v, half_t& hi, half_t& lo) {
  v = fminf(v, 65504.0f);
  hi = (half_t)v;
  lo = (half_t)(v - (float)hi);
}

template <int P, int KC, int NW, int MW, bool POOL>
__global__ __launch_bounds__(256) void conv3x3_bias_relu_kernel(ConvArgs a) {
  using C = ConvCfg<P, KC, NW, MW>;
  constexpr int TH = C::TH, TW = C::TW, HALO_W = C::HALO_W, NHALO = C::NHALO;
  constexpr int KG = C::KG, BN = C::BN, KGS = C::KGS, EPN = C::EPN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* halo = smem;
  char* slab = smem + C::HALO_BYTES;

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
  const int H = a.H, W = a.W, Cin = a.Cin;

  float16v acc[MW][NW];
#pragma unroll
  for (int m = 0; m < MW; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

  const half_t* in_n = a.in + (size_t)n * H * W * P * Cin;
  const char* wsrc = (const char*)a.wpk + (size_t)ct * a.nchunks * C::SLAB_BYTES;

  // lane-constant LDS read offsets
  const int a_lane_off = (lane >> 5) * KGS + ((wave * MW) * HALO_W + (lane & 31)) * 16;
  const int b_lane_off = ((lane >> 5) * BN + (lane & 31)) * 16;

  for (int c = 0; c < a.nchunks; ++c) {
    if (c) __syncthreads();
    // ---- stage the halo chunk: items = NHALO x P x KG of 16 bytes
    {
      constexpr int ITEMS = NHALO * P * KG;
      constexpr int ITERS = (ITEMS + 255) / 256;
      u32x4 v[ITERS];
      const int c0 = c * KC;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        int i = tid + it * 256;
        int kg = i % KG;
        int pl = (i / KG) % P;
        int hp = i / (KG * P);
        int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        int gy = y0 + hy - 1, gx = x0 + hx - 1;
        int ch = c0 + kg * 8;
        bool ok = (i < ITEMS) && (gy >= 0) && (gy < H) && (gx >= 0) && (gx < W) && (ch < Cin);
        u32x4 z = {0u, 0u, 0u, 0u};
        v[it] = z;
        if (ok) v[it] = *(const u32x4*)(in_n + ((size_t)(gy * W + gx) * P + pl) * Cin + ch);
      }
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        int i = tid + it * 256;
        int kg = i % KG;
        int pl = (i / KG) % P;
        int hp = i / (KG * P);
        if (i < ITEMS) *(u32x4*)(halo + (pl * KG + kg) * KGS + hp * 16) = v[it];
      }
    }
    // ---- stage the weight slab (linear copy)
    {
      constexpr int UNITS = C::SLAB_BYTES / 16;
      constexpr int ITERS = (UNITS + 255) / 256;
      const char* src = wsrc + (size_t)c * C::SLAB_BYTES;
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        int i = tid + it * 256;
        if (UNITS % 256 == 0 || i < UNITS) *(u32x4*)(slab + i * 16) = *(const u32x4*)(src + (size_t)i * 16);
      }
    }
    __syncthreads();
    // ---- MFMA over 9 taps x KC
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
#pragma unroll
      for (int s = 0; s < KC / 16; ++s) {
        half8 ah[MW], al[MW], bh[NW], bl[NW];
#pragma unroll
        for (int m = 0; m < MW; ++m) {
          const int off = a_lane_off + (2 * s) * KGS + ((m + dy) * HALO_W + dx) * 16;
          ah[m] = *(const half8*)(halo + off);
          if (P == 2) al[m] = *(const half8*)(halo + off + KG * KGS);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
          const int off = b_lane_off + ((tap * KG + 2 * s) * BN + j * 32) * 16;
          bh[j] = *(const half8*)(slab + off);
          if (P == 2) bl[j] = *(const half8*)(slab + off + 9 * KC * BN * 2);
        }
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            if (P == 2) {
              acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[j], acc[m][j], 0, 0, 0);
              acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[j], acc[m][j], 0, 0, 0);
            }
            acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[j], acc[m][j], 0, 0, 0);
          }
      }
    }
  }

  // ---- epilogue: scale, bias, ReLU -> LDS tile -> coalesced stores (+ fused 2x2 max-pool)
  half_t* ep = (half_t*)smem;
  const int Cout = a.Cout;
#pragma unroll
  for (int j0 = 0; j0 < NW; j0 += EPN / 32) {
    __syncthreads();
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
      half_t* out_n = a.out + (size_t)n * H * W * P * Cout;
#pragma unroll
      for (int it = 0; it < UNITS / 256; ++it) {
        int u = tid + it * 256;
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
      for (int u = tid; u < PUNITS; u += 256) {
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
