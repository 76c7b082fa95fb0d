// conv3x3_ws.h — the 3x3 convolution of conv3x3_mfma.h (every ConvBlock conv of the exact-mode UNet++, reference
// src/models/unetpp.py:13-26,68-82,104-116; first built for the narrow Cout = 32 layers), with the work of a workgroup
// SPLIT BY ROLE:
//
//   waves 0-3  consumers   fragment reads + MFMAs (one wave per SIMD, 4 rows x 32 pixels x 32 channels each), epilogue
//   waves 4-7  producers   everything that fills LDS: halo / weight-slab LDS-DMA, and (UPF) the bilinear x2 upsample
//                          of the `up` channels, interpolated into the halo image from low-res pixel records
//
// Why: with all eight waves running the same program (conv3x3_mfma.h) they move in lock-step, one workgroup per CU
// (LDS), so load issue, interpolation, matrix work and epilogue of a tile ADD UP — the Cout = 32 layers ran with the
// matrix pipe 38-52 % busy and the fused upsample's VALU work did not hide at all.  A consumer and a
// producer wave share each SIMD; the matrix pipe and the VALU/LDS/VMEM paths are separate, so the producer's
// instructions issue in the 24 of every 32 cycles an MFMA leaves free.  Producers run one K-chunk ahead: while the
// consumers multiply chunk g out of stage buffer g & 1, the producers fill buffer (g + 1) & 1; one s_barrier per
// chunk hands a buffer over in each direction.
//
// Tile = 16 rows x 32 pixels x 32 output channels, K = 9 taps x (C0 [+ C1]) in chunks of 16 channels; activations
// channel-blocked [N][C/16][H][W][P][16] (conv3x3_mfma.h); same LDS images, same packed weights.
// Consumers walk a chunk dx-major: the six halo rows a wave needs for one column shift are read once (12 ds_read_b128)
// and serve the three taps of that column (36 MFMAs): half the fragment reads of the tap-major order.
//
// UPF (fused bilinear upsample, unetpp.py:76,112-116: cat([skip, up(low)])): `in1` is the LOW-resolution tensor
// [N][C1/16][H/2][W/2][P][16].  Per tile and up-chunk the <= 10 x 18 low-res pixel records it needs arrive by LDS-DMA one
// producer iteration ahead.  With tile origins at multiples of (16, 32) every 2x2 block of halo pixels starting at an
// odd image row / column reads the SAME four low-res pixels (align_corners=True, src = dst * (in-1)/(out-1): checked
// for all sizes up to 4096 in tests/test_host_logic.py), so one producer lane takes a block x 4 channels: 8 ds_read_b64,
// the four corner values c = hi + lo (exact in fp32), two x-interpolations per row, four y-interpolations, the
// hi/lo re-split, 8 ds_write_b64.
//
// Further: HANDOFF -- layers whose producers only move data get half of every tile's epilogue from
// the consumers through LDS; a layer of two chunks and one channel tile keeps its weight slabs in LDS; the first tap
// of a tile starts from the MFMA's zero operand; C0F -- the producers compute conv0_0.conv1 themselves (below).
// Built with -fno-slp-vectorize (unet-_amd/_lib.py): packed fp32 VALU instructions are 5-10x slower beside the
// consumers' MFMA stream.  Measurement builds (-DUNETPP_WS_DBG): phase ablations and in-kernel phase stamps.
//
// EXACT8 (template parameter X8; include/unetpp.h UNETPP_PREC_EXACT8): the split product  x w = hi wh + lo wh + x wl  with
// the main term in fp16 (one v_mfma_f32_32x32x16_f16 per tap and 16 channels) and BOTH cross terms of TWO taps in one
// v_mfma_scale_f32_32x32x64_f8f6f4: per lane 32 bytes = [tap A: lo8 x 4, x8 x 4, lo8 x 4, x8 x 4 | tap B: the same] of input
// channels 8 h .. 8 h + 7 against [wh8 x 4, wl8 x 4, ...] (byte b of a lane meets byte b of the other operand's lane with the
// same l >> 5: scripts/microbench/scale_mfma_layout.hip).  Encodings and block scales: conv3x3_mfma.h (split_pack4_x8) and
// aux_kernels.h (weight_pack_x8_kernel).  The pixel records stay 64 bytes with the 8-bit planes where the fp16 lo plane was, so
// loaders, LDS images and fragment addresses are those of the two-plane format; only the consumers' chunk body, the epilogues'
// split and the producers that write activations (C0F, UPF) differ.  Logits within 5e-4 of the fp32 reference (bar: 1e-3).
//
// Split-K (ConvArgs::ksplit > 1, chosen by launch_ws_k for launches with fewer tiles than a quarter of the CUs): see the
// consumers' epilogue.
#pragma once
#include "conv3x3_mfma.h"

namespace unetpp {

// NW x MW = 32-channel blocks x rows per consumer wave: 1 x 4 (Cout = 32, 16-row tiles) or 2 x 2 (Cout = 64, 8-row
// tiles: the same 108 MFMAs per chunk and wave; the 37 KB weight slab of 64 output channels leaves LDS room for two
// 8-row halo images only).
template <int P, bool UPF, bool C0F = false, int NW_ = 1, int MW_ = 4, bool X8_ = false>
struct WsCfg {
  static constexpr bool X8 = X8_;
  static_assert(!X8 || P == 2, "EXACT8 keeps the 64-byte pixel records of the two-plane format");
  static constexpr int NT = 512, NCONS = 4, NPROD = 4, MW = MW_, NW = NW_;
  static constexpr int TH = NCONS * MW, TW = 32, HALO_W = TW + 2, NHALO = (TH + 2) * HALO_W;
  static constexpr int KC = 16, KG = 2, BN = 32 * NW;
  static_assert(MW % 2 == 0 && NW * MW == 4, "108 MFMAs per chunk and consumer wave");
  static constexpr int U = P * KG, PPP = 64 / U;
  static constexpr int HALO_PIECES = (NHALO + PPP - 1) / PPP, HALO_BYTES = HALO_PIECES * 1024;
  static constexpr int HALO_ITERS = (HALO_PIECES + NPROD - 1) / NPROD;
  // weight slab of one chunk: P planes of [tap][k-group][BN][8 halves]; X8: the hi plane, then the 8-bit weights of the
  // five tap pairs [pair][tap of the pair][h][BN][16 bytes] (EXACT8 notes in the file header)
  static constexpr int SLAB_MAIN = 9 * KC * BN * 2;
  static constexpr int SLAB_BYTES = X8 ? SLAB_MAIN + 5 * 2 * 2 * BN * 16 : P * SLAB_MAIN, SLAB_PIECES = SLAB_BYTES / 1024;
  static constexpr int SLAB_ITERS = (SLAB_PIECES + NPROD - 1) / NPROD;
  static constexpr int BUF_BYTES = HALO_BYTES + SLAB_BYTES;
  static constexpr int LSH = TH / 2 + 2, LSW = TW / 2 + 2, LS_PX = LSH * LSW, LS_REC = P * 32;
  static constexpr int LS_PIECES = (LS_PX * LS_REC + 1023) / 1024, LS_BYTES = UPF ? LS_PIECES * 1024 : 0;
  static constexpr int LS_ITERS = (LS_PIECES + NPROD - 1) / NPROD;
  // C0F (fused first ConvBlock, see the kernel): two input patches (TH+4) x (TW+4) x RGB float32 (+ 64 zero bytes)
  static constexpr int PATCH_H = TH + 4, PATCH_W = TW + 4, PATCH_FLOATS = PATCH_H * PATCH_W * 3;
  static constexpr int PATCH_BYTES = C0F ? ((PATCH_FLOATS * 4 + 64 + 255) / 256) * 256 : 0;
  static constexpr int C0_GROUPS = (NHALO + 15) / 16, C0_ITERS = (C0_GROUPS + NPROD - 1) / NPROD;
  // HANDOFF (layers whose producers only move data): each consumer wave leaves half of its finished accumulators
  // (two of its four 32x32 results, 8 KB) in LDS and the producer wave of the same number runs their epilogue while
  // the consumers already multiply the next tile -- two waves per SIMD issue the epilogue's VALU work instead of one.
  static constexpr bool HANDOFF = !UPF && !C0F;
  static constexpr int STG_WAVE = 2 * 16 * 64 * 4, STG_BYTES = HANDOFF ? NCONS * STG_WAVE : 0;
  static constexpr int LDS_BYTES = 2 * BUF_BYTES + 2 * LS_BYTES + 2 * PATCH_BYTES + STG_BYTES;
  // interpolation items: 2x2 halo blocks x channel quads
  static constexpr int BLK_Y = (TH + 2) / 2, BLK_X = HALO_W / 2, UP_ITEMS = BLK_Y * BLK_X * 4;
  static constexpr int UP_ROUNDS = (UP_ITEMS + NPROD * 64 - 1) / (NPROD * 64);
  static_assert(SLAB_BYTES % 1024 == 0 && LDS_BYTES <= 160 * 1024, "LDS budget");
};

typedef int int4v __attribute__((ext_vector_type(4)));
typedef int int8v __attribute__((ext_vector_type(8)));

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for this wave's global
// stores (vmcnt), so a consumer's epilogue stores keep draining while it already multiplies the next tile.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Epilogue of ROWS image rows (acc[m] = row gy0 + m) x 32 pixels x 32 channels [cbase, cbase + 32) held in MFMA
// accumulators (pixel on the lane; the weights are packed with rows8, so register r holds channel
// 16 * (r >> 3) + 8 * (lane >> 5) + (r & 7): eight consecutive channels of each of the tile's two records): scale, bias, ReLU,
// then either the fp16 hi/lo store (+ the fused 2x2 max-pool of rows (0,1), (2,3), ...) or the fused 1x1 head with
// softmax / argmax / class rules.  Same arithmetic, statement for statement, as the epilogue of conv3x3_bias_relu_kernel.
template <int P, int ROWS, bool POOL, bool HEAD, bool X8 = false>
__device__ __forceinline__ void ws_epilogue(const ConvArgs& a, const float16v (&acc)[ROWS], const float2* sb_lds,
                                            const float* head_lds, int n, int gy0, int x0, int cbase, int lane) {
  const int H = a.H, W = a.W;
  const int h = lane >> 5;
  const int gx = x0 + (lane & 31);
  const int nbo = a.Cout >> 4;
  float v[ROWS][16];
  float vmax = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int co = cbase + 16 * (r >> 3) + 8 * h + (r & 7);
    const float2 sb = sb_lds[co];
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      v[m][r] = fmaxf(acc[m][r] * sb.x + sb.y, 0.f);
      if (!HEAD) vmax = fmaxf(vmax, v[m][r]);           // x0_4 stays in fp32 registers in the fused head: nothing to clamp
    }
  }
  if (!HEAD && __builtin_amdgcn_ballot_w64(vmax > F16_MAX)) {      // rare: report, then clamp to what fp16 can hold
    range_flag(a.status, vmax > F16_MAX, false);
#pragma unroll
    for (int r = 0; r < 16; ++r)
#pragma unroll
      for (int m = 0; m < ROWS; ++m) v[m][r] = fminf(v[m][r], F16_MAX);
  }
  if (!HEAD) {
#pragma unroll
    for (int m = 0; m < ROWS; ++m) {
      const int gy = gy0 + m;
      const bool ok = gy < H && gx < W;
      const size_t blk = (size_t)H * W * P * 16;
      half_t* dst = a.out + ((size_t)n * nbo + (cbase >> 4)) * blk + ((size_t)gy * W + gx) * P * 16;
#ifdef UNETPP_WS_DBG
      pack_store_rows8<P, X8>(v[m], dst, blk, ok, h, (a.dbg & 262144) != 0, (a.dbg & 524288) != 0);
#else
      pack_store_rows8<P, X8>(v[m], dst, blk, ok, h);
#endif
    }
  } else {
    // logits[c] = b[c] + sum_co x0_4[co] * Wf[c][co] in fp32: a lane holds 16 of a pixel's 32 channels, lane ^ 32 the
    // other 16.  Rows are finished in pairs: one v_permlane32_swap hands the low half-wave both partial sums of row 2i
    // and the high half-wave both of row 2i+1, so each half-wave finishes (argmax, softmax, rules, stores) a different
    // row instead of both doing both.  Sum order (lanes 0-31's part first) as in conv3x3_bias_relu_kernel: same bits.
    static_assert(!HEAD || ROWS % 2 == 0, "fused head: row pairs");
    const size_t hw = (size_t)H * W;
#pragma unroll
    for (int m2 = 0; m2 < ROWS / 2; ++m2) {
      float lg[HEAD_FUSED_MAX_CLASSES];
#pragma unroll
      for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) {
        lg[c] = -INFINITY;
        if (c < a.head_C) {       // uniform branch
          float wq[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 w4 = *(const float4*)(head_lds + c * 32 + 16 * (q >> 1) + 8 * h + 4 * (q & 1));
            wq[4 * q] = w4.x; wq[4 * q + 1] = w4.y; wq[4 * q + 2] = w4.z; wq[4 * q + 3] = w4.w;
          }
          const float bc = head_lds[a.head_C * 32 + c];
          float p0 = 0.f, p1 = 0.f;
#pragma unroll
          for (int r = 0; r < 16; ++r) { p0 = fmaf(v[2 * m2][r], wq[r], p0); p1 = fmaf(v[2 * m2 + 1][r], wq[r], p1); }
          auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), false, false);
          unsigned s0 = sw[0], s1 = sw[1];
          // hipcc (ROCm 7.2) pads nothing between v_permlane32_swap and a VALU instruction reading its results, and the
          // add then sees stale registers (measured: sum = 2 * s0); stores of swapped words are not affected
          asm volatile("s_nop 1" : "+v"(s0), "+v"(s1));
          lg[c] = bc + (__builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1));
        }
      }
      const int gy = gy0 + 2 * m2 + h;                 // this half-wave's row
      const bool ok = gy < H && gx < W;
      const size_t pix = (size_t)gy * W + gx;
      float best = lg[0];
      int besti = 0;
#pragma unroll
      for (int c = 1; c < HEAD_FUSED_MAX_CLASSES; ++c)
        if (lg[c] > best) { best = lg[c]; besti = c; }        // first maximal class wins
      bool is_cable = besti == 1, is_tape = besti == 2;
      if (a.probs || a.rule) {      // uniform: softmax_np = exp(x - max) / sum, fp32
        float pe[HEAD_FUSED_MAX_CLASSES], sum = 0.f;
#pragma unroll
        for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) {
          pe[c] = (c < a.head_C) ? expf(lg[c] - best) : 0.f;
          sum += pe[c];
        }
#pragma unroll
        for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) {
          pe[c] = pe[c] / sum;
          if (ok && a.probs && c < a.head_C) a.probs[((size_t)n * a.head_C + c) * hw + pix] = pe[c];
        }
        if (a.rule) apply_rule(a.rule, pe[0], pe[1], pe[2], a.t_cable, a.t_tape, a.bg_margin, a.ct_margin, is_cable, is_tape);
      }
      if (ok) {
        if (a.logits) {
#pragma unroll
          for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c)
            if (c < a.head_C) a.logits[((size_t)n * a.head_C + c) * hw + pix] = lg[c];
        }
        const size_t o = (size_t)n * hw + pix;
        if (a.mask) a.mask[o] = (uint8_t)besti;
        if (a.cable) a.cable[o] = is_cable;
        if (a.tape) a.tape[o] = is_tape;
      }
    }
  }
  if (POOL) {
    static_assert(!POOL || ROWS % 2 == 0, "fused pool: row pairs");
    const int Hp = H >> 1, Wp = W >> 1;
#pragma unroll
    for (int m2 = 0; m2 < ROWS / 2; ++m2) {
      float pv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float vm = fmaxf(v[2 * m2][r], v[2 * m2 + 1][r]);
        const float vn = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, vm), 0xB1, 0xF, 0xF, true));
        pv[r] = fmaxf(vm, vn);       // quad_perm [1,0,3,2]: the horizontally adjacent pixel
      }
      const int py = (gy0 >> 1) + m2, px = (x0 >> 1) + ((lane & 31) >> 1);
      const bool ok = ((lane & 1) == 0) && py < Hp && px < Wp;
      const size_t blk = (size_t)Hp * Wp * P * 16;
      half_t* dst = a.pool_out + ((size_t)n * nbo + (cbase >> 4)) * blk + ((size_t)py * Wp + px) * P * 16;
#ifdef UNETPP_WS_DBG
      pack_store_rows8<P, X8>(pv, dst, blk, ok, h, (a.dbg & 262144) != 0);
#else
      pack_store_rows8<P, X8>(pv, dst, blk, ok, h);
#endif
    }
  }
}

// C0F = the whole first ConvBlock in one launch (reference unetpp.py:68,104 conv0_0 = relu(bn(conv(relu(bn(conv(x)))))),
// with preprocess_image's BGR->RGB, /255, HWC->CHW of infer_two_stage_burr.py:122-127 when the input is uint8): the
// PRODUCERS compute conv0_0.conv1 (3 -> 32 channels, K = 27) themselves, straight from the caller's input tensor, into
// the two halo images the consumers' conv0_0.conv2 reads -- x0_0a, the float32->fp16 input copy and two launches are
// gone.  Per tile: a (16+4) x (32+4) x RGB float32 patch goes to LDS (prefetched one tile ahead); per chunk (16 of the
// 32 conv1 channels) every producer wave takes 16-pixel groups of the 18 x 34 halo: it gathers each pixel's 27 patch
// values as the B operand of v_mfma_f32_16x16x32_f16 (K = 32: [dy][dx,ch] order, 5 zero slots), multiplies by the
// packed conv1 weights (hi/lo split, three MFMAs as everywhere), applies scale/bias/ReLU, writes zeros for halo pixels
// outside the image (conv2's padding) and stores the hi/lo quads into the halo image.  The two conv2 weight slabs
// stay in LDS for the whole launch.
// CAT2: the K loop runs over TWO full-resolution tensors (SimpleUNet's cat([up, enc]), simple_unet.py:112,117,122), in0's chunks
// first; both hold whole 16-channel records (the host checks), so a lane's offsets are those of the first source.
template <int P, bool POOL, bool HEAD, bool UPF, bool C0F = false, int NW = 1, int MWP = 4, bool X8 = false, bool CAT2 = false>
__global__ __launch_bounds__(512, 2) void conv3x3_ws_kernel(ConvArgs a) {
  using C = WsCfg<P, UPF, C0F, NW, MWP, X8>;
  static_assert(!C0F || (!UPF && !HEAD && P == 2 && NW == 1), "fused first block: exact mode, no other fusion in the loader");
  static_assert(!HEAD || NW == 1, "the fused head needs all 32 channels of x0_4 in one 32-block");
  constexpr int NT = C::NT, MW = C::MW, TH = C::TH, TW = C::TW, HALO_W = C::HALO_W;
  constexpr int KC = C::KC, KG = C::KG, BN = C::BN, PPP = C::PPP;
  static_assert(!(POOL && HEAD) && !(UPF && (POOL || HEAD)), "one fused extra per kernel");
  static_assert(!CAT2 || (!UPF && !C0F && !POOL && !HEAD), "two full-resolution sources: the plain kernel only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef UNETPP_WS_DBG
  const unsigned long long rt_entry = __builtin_amdgcn_s_memrealtime();
#endif
  touch_kernarg_lines<ConvArgs>();                      // (conv3x3_mfma.h: nine serialised scalar-cache misses otherwise)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;
  const int tiles_img = a.tiles_x * a.tiles_y;
  // a.nct = Cout / BN channel tiles per pixel tile; a.ksplit (>= 1) = workgroups that share a tile, each with its own
  // range of K-chunks (small batches: see the split-K notes at the consumers' epilogue)
  const int KS = a.ksplit;
  const int total_tiles = a.N * tiles_img * a.nct * KS;
  const int nch0 = (a.C0 + KC - 1) / KC;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_char_t*)smem;

  // per-channel (scale, bias) of the layer and the head weights, staged once per workgroup (read by the consumers)
  // (by the CONSUMER waves only, which have nothing else to do until the first chunk has landed: the producers go straight
  // to their first DMA instead of waiting for these loads; the first chunk's barrier publishes the table -- a launch of one
  // tile per CU, i.e. a small batch, is a chain of such latencies)
  float2* sb_lds = (float2*)(smem + C::LDS_BYTES);
  float* head_lds = (float*)(smem + C::LDS_BYTES + a.Cout * 8);
  if (wave < C::NCONS) {
    for (int i = tid; i < a.Cout; i += C::NCONS * 64) sb_lds[i] = make_float2(a.scale[i], a.bias[i]);
    if (HEAD) {
      for (int i = tid; i < a.head_C * 32; i += C::NCONS * 64) head_lds[i] = a.head_w[i];
      for (int i = tid; i < a.head_C; i += C::NCONS * 64) head_lds[a.head_C * 32 + i] = a.head_b[i];
    }
  }

#ifdef UNETPP_WS_DBG
  const unsigned long long rt_staged = __builtin_amdgcn_s_memrealtime();
#endif
  // tile order as in conv3x3_bias_relu_kernel: an XCD (workgroups b, b+8, ...) walks a contiguous run of tiles
  const int G = (int)gridDim.x;
  const int slot = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;
  // Tile number -> (channel tile, column, row, image).  A workgroup asks for its slot, then for tiles G apart (or the
  // same tile again): only the first call divides, later ones add G's own decomposition with carries (wave-uniform
  // scalar work; the six divisions per tile used to cost several hundred cycles in each role).
  int dec_ks = 0, dec_ct = 0, dec_tx = 0, dec_ty = 0, dec_n = 0, dec_t = -1;      // the tile decoded last (per-wave cursor)
  // (G's own decomposition comes from the host: five scalar divisions are ~150 dependent instructions, and at one tile
  // per workgroup -- a small batch -- the whole launch waits for the producers' path to their first DMA)
  const int g_ks = a.gdec[0], g_ct = a.gdec[1], g_tx = a.gdec[2], g_ty = a.gdec[3], g_n = a.gdec[4];
  // u = q d + r for wave-uniform 0 <= u < 2^22, d >= 1: one reciprocal and a branch-free correction step instead of the
  // generic 32-bit division (below 2^22 the float quotient is off by at most one)
  const bool small_radix = total_tiles < (1 << 22);
  auto divmod = [&](const int u, const int d, int& q, int& r) {      // (q may alias the caller's u)
    int qq, rr;
    if (small_radix) {
      qq = (int)((float)u * __builtin_amdgcn_rcpf((float)d));
      rr = u - qq * d;
      const int up = rr >= d ? 1 : 0, dn = rr < 0 ? 1 : 0;
      qq += up - dn; rr += (dn - up) * d;
    } else {
      qq = u / d; rr = u - qq * d;
    }
    q = qq; r = rr;
  };
  auto decode = [&](int t, int& n, int& y0, int& x0) {
    if (t != dec_t) {
      if (dec_t >= 0 && t == dec_t + G) {
        dec_ks += g_ks;
        int carry = dec_ks >= KS ? 1 : 0;
        dec_ks -= carry ? KS : 0;
        dec_ct += g_ct + carry;
        carry = dec_ct >= a.nct ? 1 : 0;
        dec_ct -= carry ? a.nct : 0;
        dec_tx += g_tx + carry;
        carry = dec_tx >= a.tiles_x ? 1 : 0;
        dec_tx -= carry ? a.tiles_x : 0;
        dec_ty += g_ty + carry;
        carry = dec_ty >= a.tiles_y ? 1 : 0;
        dec_ty -= carry ? a.tiles_y : 0;
        dec_n += g_n + carry;
      } else {
        int u = t;
        divmod(u, KS, u, dec_ks);
        divmod(u, a.nct, u, dec_ct);
        divmod(u, a.tiles_x, u, dec_tx);
        divmod(u, a.tiles_y, dec_n, dec_ty);
      }
      dec_t = t;
    }
    n = dec_n; x0 = dec_tx * TW; y0 = dec_ty * TH;
  };
  if (slot >= total_tiles) return;                       // whole workgroup: no barrier has been executed yet
  // (a one-chunk tile has no second barrier to publish the handed-over pair; a split tile has no epilogue to share)
  const bool handoff = C::HANDOFF && a.nchunks >= 2 && KS == 1;

  // ---- C0F: patch loader (producer lanes) and the once-per-launch part of the producers' work
  typedef __attribute__((ext_vector_type(4))) float float4v;
  const int plane_id = tid - C::NCONS * 64;              // 0..255 among the producer lanes (negative: a consumer lane)
  constexpr int PATCH_ELEMS = C::PATCH_FLOATS, PATCH_LOADS = C0F ? (PATCH_ELEMS + C::NPROD * 64 - 1) / (C::NPROD * 64) : 1;
  const int patch_base = 2 * C::BUF_BYTES + 2 * C::LS_BYTES;
  // patch element e of this lane (e = plane_id + 256 i): which image sample it is (row, col, ch relative to the patch
  // origin) and where it goes in the LDS image [row][col][RGB]; fixed for the whole launch.
  //   float32 NCHW: lanes walk a plane's rows (coalesced);  uint8 NHWC BGR: lanes walk the bytes of a row, RGB = BGR reversed
  int pe_rc[PATCH_LOADS], pe_idx[PATCH_LOADS];          // (row << 16) | (col << 2) | plane-or-byte, LDS float index or -1
  if (C0F) {
#pragma unroll
    for (int i = 0; i < PATCH_LOADS; ++i) {
      const int e = plane_id + i * (C::NPROD * 64);
      int row, col, cs, idx;
      if (a.raw_fmt == 0) {
        const int ch = e / (C::PATCH_H * C::PATCH_W), rem = e - ch * (C::PATCH_H * C::PATCH_W);
        row = rem / C::PATCH_W; col = rem - row * C::PATCH_W; cs = ch; idx = rem * 3 + ch;
      } else {
        row = e / (C::PATCH_W * 3); const int b = e - row * (C::PATCH_W * 3);
        col = b / 3; cs = b - col * 3; idx = (row * C::PATCH_W + col) * 3 + (2 - cs);
      }
      pe_rc[i] = (row << 16) | (col << 2) | cs;
      pe_idx[i] = (e < PATCH_ELEMS && plane_id >= 0) ? idx : -1;
    }
  }
  // values of the patch of tile (n, y0, x0): image pixel (y0 - 2 + row, x0 - 2 + col), 0 outside the image (buffer
  // loads: an out-of-range offset reads back 0, no branches); uint8 input: /255 (preprocess_image)
  auto patch_fetch = [&](int n, int y0, int x0, float (&val)[PATCH_LOADS]) {
    const size_t img = (size_t)H * W * 3;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const char*)a.raw_in + (size_t)n * img * (a.raw_fmt == 0 ? 4 : 1)), 0, (int)(img * (a.raw_fmt == 0 ? 4 : 1)), 0x00020000);
    bool beyond = false;
#pragma unroll
    for (int i = 0; i < PATCH_LOADS; ++i) {
      const int gy = y0 - 2 + (pe_rc[i] >> 16), gx = x0 - 2 + ((pe_rc[i] >> 2) & 0x3fff), cs = pe_rc[i] & 3;
      const bool ok = pe_idx[i] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      float v;
      if (a.raw_fmt == 0) {
        v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? ((cs * H + gy) * W + gx) * 4 : (int)0x80000000, 0, 0));
      } else {
        const unsigned char b8 = __builtin_amdgcn_raw_buffer_load_b8(rs, ok ? (gy * W + gx) * 3 + cs : (int)0x80000000, 0, 0);
        v = __fdiv_rn((float)b8, 255.0f);
      }
      beyond |= !(fabsf(v) <= F16_MAX);
      val[i] = v;
    }
    // a float32 input beyond the fp16 range (or a NaN) cannot enter the hi/lo planes unchanged: report it, clamp it
    if (__builtin_amdgcn_ballot_w64(beyond)) {
      bool nan = false;
#pragma unroll
      for (int i = 0; i < PATCH_LOADS; ++i) { nan |= val[i] != val[i]; val[i] = __builtin_amdgcn_fmed3f(val[i], -F16_MAX, F16_MAX); }
      range_flag(a.status, beyond, nan);
    }
  };
  // LDS image [row][col][RGB], one word per sample: fp16 hi in the low half, fp16 lo (= sample - hi) in the high half --
  // split once per tile here instead of once per use (every sample is an operand of up to nine pixels, in two chunks)
  auto patch_store = [&](int buf, const float (&val)[PATCH_LOADS]) {
    unsigned* pd = (unsigned*)(smem + patch_base + buf * C::PATCH_BYTES);
#pragma unroll
    for (int i = 0; i < PATCH_LOADS; ++i)
      if (pe_idx[i] >= 0) {
        unsigned wh, wl;
        split_pack2(val[i], 0.f, wh, wl);
        pd[pe_idx[i]] = (wh & 0xffffu) | (wl << 16);
      }
  };
  if (C0F && wave >= C::NCONS) {
    // conv2's two weight slabs: resident for the whole launch (buffer c holds chunk c)
    constexpr int SL_IT = (2 * C::SLAB_PIECES + C::NPROD - 1) / C::NPROD;
#pragma unroll
    for (int it = 0; it < SL_IT; ++it) {
      const int piece = (wave - C::NCONS) + it * C::NPROD;
      if (piece < 2 * C::SLAB_PIECES) {
        const int c = piece / C::SLAB_PIECES, pc = piece - c * C::SLAB_PIECES;
        glds16((const char*)a.wpk + (size_t)c * C::SLAB_BYTES + pc * 1024, lane * 16,
               lds_base + c * C::BUF_BYTES + C::HALO_BYTES + pc * 1024);
      }
    }
    // the zero words behind each patch (operand slots 27..31 of a pixel read them) and the first tile's patch
    if (plane_id < 32) {
      *(float*)(smem + patch_base + PATCH_ELEMS * 4 + (plane_id & 15) * 4 + (plane_id >> 4) * C::PATCH_BYTES) = 0.f;
    }
    int n0, y00, x00;
    decode(slot, n0, y00, x00);
    float pv[PATCH_LOADS];
    patch_fetch(n0, y00, x00, pv);
    patch_store(0, pv);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  if (C0F) __syncthreads();                              // C0F: slabs, zero words and the first patch visible to all producer waves

#ifdef UNETPP_WS_DBG
#define WS_STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); st_sum[i] += now_ - st_t; st_t = now_; \
                      if (ev_ptr && ev_ptr < ev_end) *ev_ptr++ = ((unsigned long long)(i) << 56) | __builtin_amdgcn_s_memrealtime(); }
// event log of one wave per role (dbg bit 131072, single-layer stamps only): (phase << 56 | 100 MHz wall clock) after every phase, 60 events
#define WS_EVLOG(role) unsigned long long* ev_ptr = (a.stamps && (a.dbg & 131072) && lane == 0 && blockIdx.x < 1024) ? a.stamps + (size_t)1024 * 32 + ((size_t)blockIdx.x * 2 + (role)) * 64 : nullptr; \
                       unsigned long long* const ev_end = ev_ptr + 60; if (ev_ptr) *ev_ptr++ = (63ull << 56) | rt_entry;
#define WS_STAMP_IN(i) if (a.dbg & 32768) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); WS_STAMP(i) __builtin_amdgcn_sched_barrier(0); }
#define WS_EVT(i) { if (ev_ptr && ev_ptr < ev_end) *ev_ptr++ = ((unsigned long long)(i) << 56) | __builtin_amdgcn_s_memrealtime(); }
#else
#define WS_STAMP(i) {}
#define WS_STAMP_IN(i) {}
#define WS_EVT(i) {}
#endif
  if (C0F && wave >= C::NCONS) {
    // =============================================================== producers, fused first block
    const int pw = wave - C::NCONS;
#ifdef UNETPP_WS_DBG
    // stamps: [0] patch prefetch issue, [1] conv1 groups, [2] patch store, [3] barrier
    unsigned long long st_sum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = __builtin_readcyclecounter();
    WS_EVLOG(1)
    if (pw != 0) ev_ptr = nullptr;
#endif
#ifndef UNETPP_C0_PRIO
#define UNETPP_C0_PRIO 1
#endif
    // the matrix pipe serves the older (consumer) wave first: without a higher priority the producers' few MFMAs wait
    // until the consumers stall
    if (UNETPP_C0_PRIO) __builtin_amdgcn_s_setprio(UNETPP_C0_PRIO);
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const int pxl = lane & 15, kq = lane >> 4;
    // packed conv1 weights: A fragments of this lane (row = output channel pxl of the half, k = 8 kq .. 8 kq + 7)
    half8 wa_h[2], wa_l[2];
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      wa_h[c] = *(const half8*)(a.c1w + ((size_t)(c * 2 + 0) * 64 + lane) * 8);
      wa_l[c] = *(const half8*)(a.c1w + ((size_t)(c * 2 + 1) * 64 + lane) * 8);
    }
    // scale / bias of conv1's channels 16 c + 4 kq + r
    float sc1[2][4], bi1[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc1[c][r] = a.c1_scale[16 * c + 4 * kq + r]; bi1[c][r] = a.c1_bias[16 * c + 4 * kq + r]; }
    // operand slot i of a pixel at patch position (hy, hx): byte offset relative to patch[(hy)][hx][0]
    //   kq < 3: row hy + kq, floats 0..7 of the 9 (dx, ch) values;  kq == 3: float 8 of rows hy, hy+1, hy+2, then zeros
    int koff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (kq < 3) koff[i] = (kq * C::PATCH_W * 3 + i) * 4;
      else koff[i] = i < 3 ? (i * C::PATCH_W * 3 + 8) * 4 : -1;            // -1: the zero words behind the patch
    }
    int g = 0;
    for (int tile = slot; tile < total_tiles; tile += G) {
      int n, y0, x0;
      decode(tile, n, y0, x0);
      const int tbuf = (g >> 1) & 1;                                     // patch buffer of this tile (two chunks per tile)
      const char* patch = smem + patch_base + tbuf * C::PATCH_BYTES;
      const int zero_off = PATCH_ELEMS * 4;
#pragma unroll
      for (int c = 0; c < 2; ++c, ++g) {                 // unrolled: c indexes register arrays
        // prefetch the next tile's patch into registers at the start of chunk 0, park it in LDS at the end of chunk 0
        float pv[PATCH_LOADS];
        const int ntile = tile + G;
        const bool have_next = c == 0 && ntile < total_tiles;
        if (have_next) {
          int n2, y2, x2;
          decode(ntile, n2, y2, x2);
          patch_fetch(n2, y2, x2, pv);
        }
        WS_STAMP(0)
        char* himg = smem + c * C::BUF_BYTES;
        float vmax = 0.f;
        // groups in batches of GB: all operand reads of a batch first, then its MFMAs, then its epilogues, so that the
        // LDS latency and the three dependent MFMAs of one group hide behind the other groups' work (patch reads and
        // halo-image writes go through the same LDS pointer: the compiler would not reorder them by itself)
#ifndef UNETPP_C0_GB
#define UNETPP_C0_GB 2
#endif
        constexpr int GB = UNETPP_C0_GB;
        static_assert(C::C0_ITERS % GB == 0, "group batches");
        // this lane's pixel of group pw + 4 it: hp = 16 (pw + 4 it) + pxl, advanced by 64 halo pixels per group; in the
        // halo image that is piece pw + 4 it, slot pxl: a lane-constant offset plus 4 KB per group
        const int dst_lane = ((kq >> 1) * PPP + pxl) * 16 + (kq & 1) * 8;
        auto groups = [&](auto interior_tag) {
          constexpr bool INTERIOR = decltype(interior_tag)::value;      // the whole halo lies inside the image
          int hy = (pw * 16 + pxl) / HALO_W, hx = (pw * 16 + pxl) - hy * HALO_W;
#pragma nounroll                                        // (unrolled, the per-group index math is hoisted out of the tile loop: spills)
          for (int b0 = 0; b0 < C::C0_ITERS; b0 += GB) {
            unsigned bw[GB][8];                           // packed (hi, lo) halves of the 8 operand values, see patch_store
            int ghy[GB], ghx[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
              ghy[k] = hy; ghx[k] = hx;
              const int hyc = min(hy, TH + 1);                  // the last group runs past the halo: any valid address
              const int pix_off = (hyc * C::PATCH_W + hx) * 12;
#pragma unroll
              for (int i = 0; i < 8; ++i) bw[k][i] = *(const unsigned*)(patch + (koff[i] >= 0 ? pix_off + koff[i] : zero_off + 4 * i));
              hy += 1; hx += 64 - HALO_W;                        // + 64 pixels = one row and 30 columns
              if (hx >= HALO_W) { hx -= HALO_W; hy += 1; }
            }
            f32x4 acc[GB];
#pragma unroll
            for (int k = 0; k < GB; ++k) {
              unsigned bh[4], bl[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) {               // hi halves / lo halves of two neighbouring values into one word each
                bh[i] = __builtin_amdgcn_perm(bw[k][2 * i + 1], bw[k][2 * i], 0x05040100u);
                bl[i] = __builtin_amdgcn_perm(bw[k][2 * i + 1], bw[k][2 * i], 0x07060302u);
              }
              const half8 xh = __builtin_bit_cast(half8, (u32x4){bh[0], bh[1], bh[2], bh[3]});
              const half8 xl = __builtin_bit_cast(half8, (u32x4){bl[0], bl[1], bl[2], bl[3]});
              acc[k] = (f32x4){0.f, 0.f, 0.f, 0.f};
              acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa_h[c], xl, acc[k], 0, 0, 0);
              acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa_l[c], xh, acc[k], 0, 0, 0);
              acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa_h[c], xh, acc[k], 0, 0, 0);
            }
#pragma unroll
            for (int k = 0; k < GB; ++k) {
              // lane: pixel pxl of the group, channels 16 c + 4 kq + r.  Halo pixels outside the image are conv2's zero padding.
              const int gi = pw + (b0 + k) * C::NPROD;           // group index = piece of the halo image
              const bool valid = gi * 16 + pxl < C::NHALO;
              float v[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = acc[k][r] * sc1[c][r] + bi1[c][r];
              vmax = fmaxf(fmaxf(vmax, v[0]), v[1]); vmax = fmaxf(fmaxf(vmax, v[2]), v[3]);          // v_max3_f32; before the clamp
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = __builtin_amdgcn_fmed3f(v[r], 0.f, F16_MAX);        // ReLU and the fp16 range in one
              if (!INTERIOR) {
                const int gy = y0 - 1 + ghy[k], gx = x0 - 1 + ghx[k];
                const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = inside ? v[r] : 0.f;
              }
              unsigned oh0, oh1, ol0, ol1;
              if (X8) {
                split_pack4_x8(v[0], v[1], v[2], v[3], oh0, oh1, ol0, ol1);      // {lo8 x 4, x8 x 4} where the lo quad used to go
              } else {
                split_pack2(v[0], v[1], oh0, ol0);
                split_pack2(v[2], v[3], oh1, ol1);
              }
              if (valid) {
                char* dst = himg + gi * 1024 + dst_lane;
                *(u32x2*)dst = (u32x2){oh0, oh1};
                *(u32x2*)(dst + KG * PPP * 16) = (u32x2){ol0, ol1};
              }
            }
          }
        };
        if (y0 >= 1 && y0 + TH < H && x0 >= 1 && x0 + TW < W) groups(std::true_type());
        else groups(std::false_type());
        if (__builtin_amdgcn_ballot_w64(vmax > F16_MAX)) range_flag(a.status, vmax > F16_MAX, false);
        WS_STAMP(1)
        if (have_next) patch_store(tbuf ^ 1, pv);
        WS_STAMP(2)
        lds_barrier();                                    // chunk g published; the consumers have left the other image
        WS_STAMP(3)
      }
    }
#ifdef UNETPP_WS_DBG
    if (a.stamps && pw == 0 && lane == 0)
#pragma unroll
      for (int i = 0; i < 11; ++i) a.stamps[((size_t)blockIdx.x * 2 + 1) * 16 + i] = st_sum[i];
    if (a.stamps && pw == 0 && lane == 0) a.stamps[((size_t)blockIdx.x * 2 + 1) * 16 + 12] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }

  if (wave >= C::NCONS) {
    // =============================================================== producers
    const int pw = wave - C::NCONS;
#ifdef UNETPP_WS_DBG
    const unsigned long long rt_prod = __builtin_amdgcn_s_memrealtime();
    { const int pr = (a.dbg >> 10) & 3; if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 3) __builtin_amdgcn_s_setprio(3); }
#endif
    constexpr unsigned OOB = 0x80000000u;                // beyond num_records: the buffer load returns zeros
    constexpr int ITERS = C::HALO_ITERS;
    // tile-invariant: halo pixel / unit of this lane in each of its DMA pieces (as in conv3x3_bias_relu_kernel)
    int my_u = lane / PPP;
#ifdef UNETPP_WS_DBG
    if (a.dbg & 16384) my_u = lane % C::U;               // timing experiment: a pixel's units on neighbouring lanes (results garbage)
#endif
    const int my_pl = my_u / KG, my_k8 = (my_u % KG) * 8;
    int hyx[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      int hp = (pw + it * C::NPROD) * PPP + lane % PPP;
#ifdef UNETPP_WS_DBG
      if (a.dbg & 16384) hp = (pw + it * C::NPROD) * PPP + lane / C::U;
#endif
      const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
      hyx[it] = (hp < C::NHALO) ? ((hy << 8) | hx) : -1;
    }
    const int cb0 = a.C0 < 16 ? a.C0 : 16;
    const unsigned plane_bytes0 = (unsigned)H * (unsigned)W * (unsigned)(P * cb0 * 2);      // < 2^31: unetpp_create's shape limit
    const unsigned img_bytes0 = (unsigned)H * (unsigned)W * (unsigned)(P * a.C0 * 2);
    const int Hs = H >> 1, Ws = W >> 1;
    const unsigned plane_bytes1 = (unsigned)Hs * (unsigned)Ws * (unsigned)(P * 16 * 2);
    const unsigned img_bytes1 = (unsigned)Hs * (unsigned)Ws * (unsigned)(P * a.C1 * 2);
    const float up_sh = Hs > 1 ? (float)(Hs - 1) / (float)(H - 1) : 0.f;
    const float up_sw = Ws > 1 ? (float)(Ws - 1) / (float)(W - 1) : 0.f;

    // Tile-invariant part of a lane's halo offsets: a tile whose halo lies inside the image (most of them) only adds
    // its origin; the per-pixel bounds tests are left to the tiles on the image border.
    int lane_off0[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int hy = hyx[it] >> 8, hx = hyx[it] & 255;
      const bool valid = hyx[it] >= 0 && my_k8 < a.C0;
      lane_off0[it] = valid ? (my_k8 >> 4) * (int)plane_bytes0 + ((((hy - 1) * W + (hx - 1)) * P + my_pl) * cb0 + (my_k8 & 15)) * 2 : (int)OOB;
    }
    unsigned voff0[ITERS];
    __amdgpu_buffer_rsrc_t rsrc0, rsrc1;
    constexpr int LSR = UPF ? C::LS_ITERS : 1, UPR = UPF ? C::UP_ROUNDS : 1;
    unsigned voffL[LSR];
    // interpolation item r of this lane: block (by, bx) x channel quad q
    int up_o00[UPR], up_o01[UPR], up_o10[UPR];          // staging offsets of the corners (ra,ca), (ra,cb), (rb,ca)
    float up_wx[UPR][4], up_wy[UPR][4];                  // {lx0, lx1} of the block's two columns, {ly0, ly1} of its two rows
    int up_st[UPR][4];                                   // halo-image byte offsets of the block's four pixels (2 ky + kx), [0] = -1: no item
    // A corner is only ever used as hi + lo, so the two 8-byte reads of a record may come in either order: the second
    // 16 lanes of every 32-lane read group (blocks 4..7 of eight neighbouring blocks) fetch lo first.  With 64-byte
    // records the first read then touches banks [16 rx, 16 rx + 8) in blocks 0..3 and [16 rx + 8, 16 rx + 16) in
    // blocks 4..7 -- all 64 banks once, instead of every bank twice; two separate ds_read_b64 (2 LDS cycles each)
    // also replace the ds_read2_b64 the compiler forms from a fixed +32 offset (8 cycles, banked modulo 32).
    int rd_swap = (P == 2 && !X8 && (lane & 16)) ? 32 : 0;      // (X8: hi and the 8-bit planes are not interchangeable)
#ifdef UNETPP_WS_DBG
    if (a.dbg & 32) rd_swap = 0;
#endif
    if (UPF) {
#pragma unroll
      for (int r = 0; r < UPR; ++r) {
        const int i = pw * 64 + lane + r * (C::NPROD * 64);
        const int q = i & 3, blk = i >> 2;
        const int by = blk / C::BLK_X, bx = blk - by * C::BLK_X;
        const int hp0 = (2 * by) * HALO_W + 2 * bx;       // first pixel of the block; right neighbour hp0 + 1, the row below hp0 + HALO_W
        // byte offset of channel quad q of halo pixel hp, plane 0: (hp / PPP) * 1024 + ((q >> 1) * PPP + hp % PPP) * 16 + (q & 1) * 8.
        // Lanes of the second channel octet (q >= 2) take the block's two columns in the opposite order (see setup_tile).
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const int hp = hp0 + (px >> 1) * HALO_W + ((px & 1) ^ ((q >> 1) & 1));
          up_st[r][px] = (hp / PPP) * 1024 + ((q >> 1) * PPP + hp % PPP) * 16 + (q & 1) * 8;
        }
        if (i >= C::UP_ITEMS) up_st[r][0] = -1;
      }
    }

    auto setup_tile = [&](int n, int y0, int x0) {
      if (y0 >= 1 && y0 + TH < H && x0 >= 1 && x0 + TW < W) {          // halo rows y0-1 .. y0+TH, columns x0-1 .. x0+TW
        const unsigned origin = ((unsigned)y0 * (unsigned)W + (unsigned)x0) * (unsigned)(P * cb0 * 2);
#pragma unroll
        for (int it = 0; it < ITERS; ++it) voff0[it] = origin + (unsigned)lane_off0[it];   // OOB + origin stays out of range
      } else {
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
          const int hy = hyx[it] >> 8, hx = hyx[it] & 255;
          const int gy = y0 + hy - 1, gx = x0 + hx - 1;
          const bool ok = hyx[it] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
          const unsigned pix = (unsigned)((gy * W + gx) * P + my_pl);
          voff0[it] = (ok && my_k8 < a.C0) ? (unsigned)(my_k8 >> 4) * plane_bytes0 + (pix * cb0 + (my_k8 & 15)) * 2u : OOB;
        }
      }
      rsrc0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in0 + (size_t)n * H * W * P * a.C0), 0, (int)img_bytes0, 0x00020000);
      if (CAT2)
        rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in1 + (size_t)n * H * W * P * a.C1), 0, (int)((unsigned)H * (unsigned)W * (unsigned)(P * a.C1 * 2)), 0x00020000);
      if (UPF) {
        rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in1 + (size_t)n * Hs * Ws * P * a.C1), 0, (int)img_bytes1, 0x00020000);
        const int ybase = (int)(up_sh * (float)max(y0 - 1, 0)), xbase = (int)(up_sw * (float)max(x0 - 1, 0));
        constexpr int PXP = 1024 / C::LS_REC, PARTS = C::LS_REC / 16;
#pragma unroll
        for (int it = 0; it < LSR; ++it) {
          const int lp = (pw + it * C::NPROD) * PXP + lane / PARTS;
          const int ly = lp / C::LSW, lx = lp - ly * C::LSW;
          const int yy = min(ybase + ly, Hs - 1), xx = min(xbase + lx, Ws - 1);
          voffL[it] = lp < C::LS_PX ? (unsigned)((yy * Ws + xx) * C::LS_REC + (lane % PARTS) * 16) : OOB;
        }
#pragma unroll
        for (int r = 0; r < UPR; ++r) {
          const int i = pw * 64 + lane + r * (C::NPROD * 64);
          const int q = i & 3, blk = i >> 2;
          const int by = blk / C::BLK_X, bx = blk - by * C::BLK_X;
          // the block's two image rows / columns: the first is odd (or -1), the second even; those inside the image
          // share their low-res corner pair (see the header), taken from the second unless only the first is inside
          const int gyA = y0 + 2 * by - 1, gxA = x0 + 2 * bx - 1;
          const int gyR = (gyA + 1 < H) ? gyA + 1 : gyA, gxR = (gxA + 1 < W) ? gxA + 1 : gxA;
          const float fyR = up_sh * (float)max(gyR, 0), fxR = up_sw * (float)max(gxR, 0);
          const int yy0 = min((int)fyR, Hs - 1), xx0 = min((int)fxR, Ws - 1);
          const int yy1 = yy0 + (yy0 < Hs - 1 ? 1 : 0), xx1 = xx0 + (xx0 < Ws - 1 ? 1 : 0);
          const int ry0 = min(max(yy0 - ybase, 0), C::LSH - 1), ry1 = min(max(yy1 - ybase, 0), C::LSH - 1);
          const int rx0 = min(max(xx0 - xbase, 0), C::LSW - 1), rx1 = min(max(xx1 - xbase, 0), C::LSW - 1);
          up_o00[r] = (ry0 * C::LSW + rx0) * C::LS_REC + q * 8 + rd_swap;
          up_o01[r] = (ry0 * C::LSW + rx1) * C::LS_REC + q * 8 + rd_swap;
          up_o10[r] = (ry1 * C::LSW + rx0) * C::LS_REC + q * 8 + rd_swap;
          // Lanes of the second channel octet (q >= 2) take the block's two columns in the opposite order: at every
          // ds_write_b64 the two octets of a block then hit different pixels, hence different banks (conflict-free
          // stores into the [unit][pixel] halo image; in the same order the octets would collide 2-way).
          const int xs = (q >> 1) & 1;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int gy = gyA + k, gx = gxA + (k ^ xs);
            const bool iny = gy >= 0 && gy < H, inx = gx >= 0 && gx < W;
            const float ly1 = fminf(fmaxf(up_sh * (float)max(gy, 0) - (float)yy0, 0.f), 1.f);
            const float lx1 = fminf(fmaxf(up_sw * (float)max(gx, 0) - (float)xx0, 0.f), 1.f);
            up_wy[r][2 * k] = iny ? 1.f - ly1 : 0.f; up_wy[r][2 * k + 1] = iny ? ly1 : 0.f;   // zero padding: all weights 0
            up_wx[r][2 * k] = inx ? 1.f - lx1 : 0.f; up_wx[r][2 * k + 1] = inx ? lx1 : 0.f;
          }
        }
      }
    };

#ifdef UNETPP_WS_DBG
    // stamps (macros near the top of the kernel): [0] skip-chunk issue, [1] skip-chunk wait, [2] skip-chunk barrier, [3] up-chunk issue, [4] interpolation,
    // [5] up-chunk wait, [6] up-chunk barrier, [7] tile setup; with dbg bit 32768 the interpolation is cut into
    // [8] corner reads issued and landed, [9] arithmetic, [10] stores issued and retired (instead of [4])
    unsigned long long st_sum[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_t = 0;
    WS_EVLOG(1)
    if (pw != 0) ev_ptr = nullptr;
    if (ev_ptr) { *ev_ptr++ = (48ull << 56) | rt_staged; *ev_ptr++ = (49ull << 56) | rt_prod; }
    WS_EVT(50)
#endif
    // all interpolation items of this lane for up-chunk c: staging buffer (c & 1) -> halo image.  Straight-line code
    // for all rounds (the LDS reads of every round are issued before the first value is needed; a lane without an
    // item in the last round computes on clamped addresses and skips only the stores).
    auto up_rounds = [&](int c, int halo_off) {
      typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
      const char* ls = smem + 2 * C::BUF_BYTES + (c & 1) * C::LS_BYTES;
      u32x2 hq[UPR][4], lq[UPR][4];                      // corners (ra,ca), (ra,cb), (rb,ca), (rb,cb): 4 channels each
#pragma unroll
      for (int r = 0; r < UPR; ++r) {
        const int o11 = up_o10[r] + (up_o01[r] - up_o00[r]);
        const int off[4] = {up_o00[r], up_o01[r], up_o10[r], o11};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
#ifdef UNETPP_WS_DBG
          if (a.dbg & 512) { hq[r][k] = (u32x2){0x3c003c00u, 0x3c003c00u}; lq[r][k] = (u32x2){0u, 0u}; continue; }
#endif
          hq[r][k] = *(const u32x2*)(ls + off[k]);
          if (P == 2) lq[r][k] = *(const u32x2*)(ls + (off[k] ^ 32));
        }
      }
      // LDS operations retire in order: a corner read issued behind a round's stores would only return after them
      // (stores queue behind the consumers' fragment reads for hundreds of cycles), so every read goes out first.
      __builtin_amdgcn_sched_barrier(0);
      WS_STAMP_IN(8)
#pragma unroll
      for (int r = 0; r < UPR; ++r) {
        float v[4][4];                                   // [pixel 2 * ky + kx][channel]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float cc[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            // corner value = hi + lo, exact in fp32; one mixed-precision FMA reads both halves out of the packed words
            if (X8) {
              // value = hi + 2^-8 lo8: the low-res tensor is what the previous layer stored (hi, lo8); word 0 of the second read
              // holds the four lo8 bytes of this channel quad.  One v_cvt_scalef32_pk_f16_bf8 turns a byte pair into a packed
              // fp16 pair (exact: three significant bits, 2^-24 the smallest), then the same mixed-precision FMA as in `exact`.
              const unsigned lo16 = __builtin_bit_cast(unsigned, (e >> 1) ? __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(lq[r][k][0], X8_LO_MUL, true)
                                                                        : __builtin_amdgcn_cvt_scalef32_pk_f16_bf8(lq[r][k][0], X8_LO_MUL, false));
              if (e & 1) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(cc[k]) : "v"(hq[r][k][e >> 1]), "v"(lo16));
              else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(cc[k]) : "v"(hq[r][k][e >> 1]), "v"(lo16));
            } else if (P == 2) {
              if (e & 1) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(cc[k]) : "v"(hq[r][k][e >> 1]), "v"(lq[r][k][e >> 1]));
              else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(cc[k]) : "v"(hq[r][k][e >> 1]), "v"(lq[r][k][e >> 1]));
            } else {
              const unsigned hw = hq[r][k][e >> 1];      // (a scalar first: bit-casting a vector element reads element 0)
              const half2v hv = __builtin_bit_cast(half2v, hw);
              cc[k] = (float)hv[e & 1];
            }
          }
#pragma unroll
          for (int kx = 0; kx < 2; ++kx) {
            const float t0 = fmaf(up_wx[r][2 * kx + 1], cc[1], up_wx[r][2 * kx] * cc[0]);     // x inside each row first
            const float t1 = fmaf(up_wx[r][2 * kx + 1], cc[3], up_wx[r][2 * kx] * cc[2]);
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) v[2 * ky + kx][e] = fmaf(up_wy[r][2 * ky + 1], t1, up_wy[r][2 * ky] * t0);
          }
        }
        WS_STAMP_IN(9)
#ifdef UNETPP_WS_DBG
        if (a.dbg & 64) { if (v[0][0] == 12345.f && v[3][3] == 7.f && v[1][2] == 3.f && v[2][1] == 9.f) a.status[1] = 1; continue; }
#endif
        if (up_st[r][0] >= 0) {
#pragma unroll
          for (int ky = 0; ky < 2; ++ky)
#pragma unroll
            for (int kx = 0; kx < 2; ++kx) {
              const int px = 2 * ky + kx;
              unsigned h0, h1, l0 = 0, l1 = 0;
              if (X8) {
                split_pack4_x8(v[px][0], v[px][1], v[px][2], v[px][3], h0, h1, l0, l1);
              } else if (P == 2) {
                split_pack2(v[px][0], v[px][1], h0, l0);
                split_pack2(v[px][2], v[px][3], h1, l1);
              } else {
                half2v a0 = {(half_t)v[px][0], (half_t)v[px][1]}, a1 = {(half_t)v[px][2], (half_t)v[px][3]};
                h0 = __builtin_bit_cast(unsigned, a0); h1 = __builtin_bit_cast(unsigned, a1);
              }
              const u32x2 oh = {h0, h1}, ol = {l0, l1};
              char* dst = smem + halo_off + up_st[r][px];
              *(u32x2*)dst = oh;
              if (P == 2) *(u32x2*)(dst + KG * PPP * 16) = ol;
            }
        }
        WS_STAMP_IN(10)
      }
    };

    // HANDOFF: the epilogue of the accumulator pair consumer wave `pw` left in LDS for the previous tile
    const int stg_base = 2 * C::BUF_BYTES + 2 * C::LS_BYTES + 2 * C::PATCH_BYTES;
    int hn = 0, hy0 = 0, hx0 = 0, hct = 0;
    bool have_handoff = false;
    auto handoff_epilogue = [&]() {
      const char* st = smem + stg_base + pw * C::STG_WAVE + lane * 16;
      float16v pair[2];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          typedef __attribute__((ext_vector_type(4))) float f32x4;
          const f32x4 v4 = *(const f32x4*)(st + (i * 4 + q) * 1024);
          pair[i][4 * q] = v4[0]; pair[i][4 * q + 1] = v4[1]; pair[i][4 * q + 2] = v4[2]; pair[i][4 * q + 3] = v4[3];
        }
      // the pair is rows (2, 3) of the wave's four rows (MW = 4) or the second 32-channel block of its two rows (MW = 2)
      const int gy0 = hy0 + pw * MW + (MW == 4 ? 2 : 0);
      const int cbase = hct * BN + (MW == 4 ? 0 : 32);
      ws_epilogue<P, 2, POOL, HEAD, X8>(a, pair, sb_lds, head_lds, hn, gy0, hx0, cbase, lane);
    };
    int g = 0;                                           // global chunk counter: chunk g -> stage buffer g & 1
    const bool slabs_stay = a.nchunks == 2 && a.nct == 1 && KS == 1;
    const int chunks_per_share = KS == 1 ? a.nchunks : a.nchunks / KS;
#ifdef UNETPP_WS_DBG
    st_t = __builtin_readcyclecounter();
#endif
    WS_EVT(51)
    for (int tile = slot; tile < total_tiles; tile += G) {
      int n, y0, x0;
      decode(tile, n, y0, x0);
      WS_EVT(52)
      const char* wsrc = (const char*)a.wpk + (size_t)dec_ct * a.nchunks * C::SLAB_BYTES;
      const int tile_ct = dec_ct;
      const int c_begin = dec_ks * chunks_per_share, c_end = c_begin + chunks_per_share;       // this workgroup's share of K
      setup_tile(n, y0, x0);
      WS_STAMP(7)
      for (int c = c_begin; c < c_end; ++c, ++g) {
        const int buf = (g & 1) * C::BUF_BYTES;
#ifdef UNETPP_WS_DBG
        if (!(a.dbg & 2))
#endif
        if (c < nch0) {
#pragma unroll
          for (int it = 0; it < ITERS; ++it) {
            const int piece = pw + it * C::NPROD;
            if (C::HALO_PIECES % C::NPROD == 0 || piece < C::HALO_PIECES)
              blds16<true>(rsrc0, voff0[it], c * (int)plane_bytes0, lds_base + buf + piece * 1024);
          }
        } else if (CAT2) {                               // chunk of the second full-resolution source
#pragma unroll
          for (int it = 0; it < ITERS; ++it) {
            const int piece = pw + it * C::NPROD;
            if (C::HALO_PIECES % C::NPROD == 0 || piece < C::HALO_PIECES)
              blds16<true>(rsrc1, voff0[it], (c - nch0) * (int)plane_bytes0, lds_base + buf + piece * 1024);
          }
        }
        // A layer of two chunks and one channel tile keeps its weights: chunk c of every tile lands in stage c, so the
        // slabs this workgroup loaded for its first tile are still there (a third of the bytes and DMA pieces saved).
#ifdef UNETPP_WS_DBG
        if (!(a.dbg & 8))
#endif
        if (!(slabs_stay && g >= 2))
#pragma unroll
        for (int it = 0; it < C::SLAB_ITERS; ++it) {
          const int piece = pw + it * C::NPROD;
          if (C::SLAB_PIECES % C::NPROD == 0 || piece < C::SLAB_PIECES)
            glds16(wsrc + (size_t)c * C::SLAB_BYTES + piece * 1024, lane * 16, lds_base + buf + C::HALO_BYTES + piece * 1024);
        }
        if (!UPF) { WS_STAMP(0) }
        if (UPF) {
#ifdef UNETPP_WS_DBG
          if (!(a.dbg & 2))
#endif
          if (c + 1 < a.nchunks && c + 1 >= nch0) {      // low-res records of the NEXT chunk, one iteration ahead
#pragma unroll
            for (int it = 0; it < LSR; ++it) {
              const int piece = pw + it * C::NPROD;
              if (C::LS_PIECES % C::NPROD == 0 || piece < C::LS_PIECES)
                blds16(rsrc1, voffL[it], (c + 1 - nch0) * (int)plane_bytes1,
                       lds_base + 2 * C::BUF_BYTES + ((c + 1) & 1) * C::LS_BYTES + piece * 1024);
            }
          }
          if (c >= nch0) { WS_STAMP(3) } else { WS_STAMP(0) }
#ifdef UNETPP_WS_DBG
          if (!(a.dbg & 1))
#endif
          if (c >= nch0) up_rounds(c, buf);              // this chunk's halo image from the records that landed last iteration
          if (c >= nch0) { WS_STAMP(4) } else { WS_STAMP(0) }
        }
#ifdef UNETPP_WS_DBG
        if (!(a.dbg & 128))
#endif
        if (handoff && c == 1 && have_handoff) handoff_epilogue();     // previous tile's pair: published by the barrier behind chunk 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed
        if (UPF && c >= nch0) { WS_STAMP(5) } else { WS_STAMP(1) }
        lds_barrier();                                    // chunk g published; the consumers have left buffer (g+1) & 1
        if (UPF && c >= nch0) { WS_STAMP(6) } else { WS_STAMP(2) }
      }
      hn = n; hy0 = y0; hx0 = x0; hct = tile_ct; have_handoff = true;
    }
    if (handoff) {
      lds_barrier();                                      // the consumers' last pair is in LDS
      if (have_handoff) handoff_epilogue();
    }
#ifdef UNETPP_WS_DBG
    if (a.stamps && pw == 0 && lane == 0)
#pragma unroll
      for (int i = 0; i < 11; ++i) a.stamps[((size_t)blockIdx.x * 2 + 1) * 16 + i] = st_sum[i];
    if (a.stamps && pw == 0 && lane == 0) a.stamps[((size_t)blockIdx.x * 2 + 1) * 16 + 12] = __builtin_amdgcn_s_memrealtime();
#endif
    return;
  }

  // ================================================================= consumers
  const int cw = wave;
#ifdef UNETPP_WS_DBG
  { const int pr = (a.dbg >> 12) & 3; if (pr == 1) __builtin_amdgcn_s_setprio(1); else if (pr == 2) __builtin_amdgcn_s_setprio(2); else if (pr == 3) __builtin_amdgcn_s_setprio(3); }
#endif
  // lane-constant LDS read offsets: pixel fragment of halo row r (0 .. MW+1 of this wave) shifted by dx
  int a_off[MW + 2][3];
#pragma unroll
  for (int r = 0; r < MW + 2; ++r)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hp = (cw * MW + r) * HALO_W + (lane & 31) + dx;
      a_off[r][dx] = (hp / PPP) * 1024 + ((lane >> 5) * PPP + hp % PPP) * 16;
    }
  const int b_lane_off = ((lane >> 5) * BN + (lane & 31)) * 16;
  struct AFrag { half8 h[MW + 2], l[MW + 2]; };
  struct BFrag { half8 h[NW], l[NW]; };
  auto load_a = [&](AFrag& f, const char* halo, int dx) {
#pragma unroll
    for (int r = 0; r < MW + 2; ++r) {
      const char* p = halo + a_off[r][dx];
      f.h[r] = *(const half8*)p;
      if (P == 2) f.l[r] = *(const half8*)(p + KG * PPP * 16);
    }
  };
  auto load_a_part = [&](AFrag& f, const char* halo, int dx, int r0, int r1) {
#pragma unroll
    for (int r = 0; r < MW + 2; ++r)
      if (r >= r0 && r < r1) {
        const char* p = halo + a_off[r][dx];
        f.h[r] = *(const half8*)p;
        if (P == 2) f.l[r] = *(const half8*)(p + KG * PPP * 16);
      }
  };
  auto load_b = [&](BFrag& f, const char* slab, int tap) {
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int off = b_lane_off + (tap * KG * BN + j * 32) * 16;
      f.h[j] = *(const half8*)(slab + off);
      if (P == 2) f.l[j] = *(const half8*)(slab + off + 9 * KC * BN * 2);
    }
  };
  float16v acc[MW][NW];
  auto run_mfma = [&](const AFrag& fa, const BFrag& fb, int dy) {
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        if (P == 2) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.h[j], fa.l[m + dy], acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.l[j], fa.h[m + dy], acc[m][j], 0, 0, 0);
        }
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.h[j], fa.h[m + dy], acc[m][j], 0, 0, 0);
      }
  };
  // the first tap of a tile's first chunk starts every accumulator from the constant 0 (an inline operand of the
  // MFMA): a tile without start values never writes zeros into its 64 accumulator registers
  auto run_mfma_first = [&](const AFrag& fa, const BFrag& fb) {
    const float16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        if (P == 2) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.h[j], fa.l[m], zero, 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.l[j], fa.h[m], acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.h[j], fa.h[m], acc[m][j], 0, 0, 0);
        } else {
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb.h[j], fa.h[m], zero, 0, 0, 0);
        }
      }
  };

  int g = 0;
#ifdef UNETPP_WS_DBG
  unsigned long long st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // [0] barrier, [1] chunk, [2] epilogue, [3] accumulator init
  unsigned long long st_t = __builtin_readcyclecounter();
  const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();     // 100 MHz, one clock for the whole chip
  WS_EVLOG(0)
  if (cw != 0) ev_ptr = nullptr;
  if (ev_ptr) *ev_ptr++ = (62ull << 56) | rt_begin;
#endif
  for (int tile = slot; tile < total_tiles; tile += G) {
    int n, y0, x0;
    decode(tile, n, y0, x0);
    const int ct = dec_ct, ks = dec_ks;
    const int c_begin = ks * (a.nchunks / KS), c_end = c_begin + a.nchunks / KS;
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        if (a.zinit && ks == 0) {    // accumulators start from the low-resolution half of the layer (tapmm_ws.h); uniform branch
          typedef __attribute__((ext_vector_type(4))) float f32x4;
          const int gy = y0 + cw * MW + m, gx = x0 + (lane & 31);
          const bool in = gy < H && gx < W;
          const float* zp = a.zinit + (((size_t)n * (a.Cout >> 5) + ((ct * BN + j * 32) >> 5)) * ((size_t)H * W) + (size_t)gy * W + gx) * 32 + 8 * (lane >> 5);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            if (in) z4 = *(const f32x4*)(zp + 16 * (q >> 1) + 4 * (q & 1));     // registers 4 q .. 4 q + 3 (rows8 channel order)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][j][4 * q + i] = z4[i];
          }
        }
      }
    const bool from_zero = a.zinit == nullptr || ks != 0;
    int4v x9[MW];                                           // EXACT8: the ninth tap's 8-bit operand of the previous chunk, see there
    if (X8) {
#pragma unroll
      for (int m = 0; m < MW; ++m) x9[m] = (int4v){0, 0, 0, 0};     // (finite bytes: a tile's first chunk meets zero weights there)
    }
    WS_STAMP(3)
    for (int c = c_begin; c < c_end; ++c, ++g) {
      lds_barrier();                                      // chunk g is in stage buffer g & 1
      WS_STAMP(0)
      const char* halo = smem + (g & 1) * C::BUF_BYTES;
      const char* slab = halo + C::HALO_BYTES;
      // dx-major walk: the six halo rows of one column shift serve its three taps.  Fragments of the next tap
      // (and, spread over the three taps, the next column shift) are read before this tap's MFMAs.
#ifdef UNETPP_WS_DBG
      if (a.dbg & 16) continue;
#endif
      if constexpr (X8) {
        // ---- EXACT8 chunk: 9 main-term steps (fp16 hi planes, one 32-cycle MFMA per product) and 5 cross-term steps (one
        // 64-cycle K = 64 MFMA per product and tap PAIR).  Order: column 0, column 1, pairs 0-2 (the taps of columns 0 and 1
        // row by row), column 2, pairs 3-4 (column 2: dy 0 + dy 1, dy 2 + a zero-weight slot).  Each phase reads the
        // fragments of the next one: h planes of a column during the phase before it, the 8-bit operands of pairs 0-2 during
        // column 1, those of pairs 3-4 during column 2.
        constexpr int R = MW + 2;
        const char* xslab = slab + C::SLAB_MAIN;
        struct BM { half8 w[NW]; };
        struct BX { int8v w[NW]; };
        auto load_h = [&](half8 (&f)[R], int dx, int r0, int r1) {
#pragma unroll
          for (int r = 0; r < R; ++r)
            if (r >= r0 && r < r1) f[r] = *(const half8*)(halo + a_off[r][dx]);
        };
        // 8-bit operand of one pixel row for a tap pair: bytes 0-15 from halo row ra shifted by dxa, bytes 16-31 from (rb, dxb)
        auto load_x = [&](int dxa, int ra, int dxb, int rb) {
          const int4v lo = *(const int4v*)(halo + a_off[ra][dxa] + KG * PPP * 16);
          const int4v hi = *(const int4v*)(halo + a_off[rb][dxb] + KG * PPP * 16);
          return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        };
        auto load_bm = [&](BM& f, int tap) {
#pragma unroll
          for (int j = 0; j < NW; ++j) f.w[j] = *(const half8*)(slab + b_lane_off + (tap * KG * BN + j * 32) * 16);
        };
        auto load_bx = [&](BX& f, int pair) {
#pragma unroll
          for (int j = 0; j < NW; ++j) {
            const char* p = xslab + pair * (2 * 2 * BN * 16) + b_lane_off + j * 32 * 16;
            const int4v lo = *(const int4v*)p, hi = *(const int4v*)(p + 2 * BN * 16);
            f.w[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
          }
        };
        auto main_step = [&](const BM& b, const half8 (&hf)[R], int dy, bool first) {
#ifdef UNETPP_WS_DBG
          if (a.dbg & 4) return;
#endif
          const float16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int j = 0; j < NW; ++j)
              acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b.w[j], hf[m + dy], first ? zero : acc[m][j], 0, 0, 0);
        };
        auto cross_step = [&](const BX& b, const int8v (&x)[R], int r0) {
#ifdef UNETPP_WS_DBG
          if (a.dbg & (4 | 65536)) return;      // 65536: the cross terms only (how much of a chunk they are)
#endif
#pragma unroll
          for (int m = 0; m < MW; ++m)
#pragma unroll
            for (int j = 0; j < NW; ++j)
              acc[m][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b.w[j], x[m + r0], acc[m][j], 0 /* e4m3 */, 1 /* e5m2 */,
                                                                         0, X8_SCALE_W, 0, X8_SCALE_A);
        };
        half8 h0[R], h1[R];
        int8v xa[R];
        BM bm0, bm1;
        BX bx0, bx1;
        load_h(h0, 0, 0, R);
        load_bm(bm0, 0);
        const bool first = c == c_begin && from_zero;
        // column 0 (taps dy * 3 + 0)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          BM& b = (dy & 1) ? bm1 : bm0;
          BM& bn = (dy & 1) ? bm0 : bm1;
          load_bm(bn, dy < 2 ? (dy + 1) * 3 : 1);
          load_h(h1, 1, 2 * dy, 2 * dy + 2);
          __builtin_amdgcn_sched_barrier(0);
          if (dy == 0 && first) main_step(b, h0, 0, true); else main_step(b, h0, dy, false);
          __builtin_amdgcn_sched_barrier(0);
        }
        // column 1 (steps 3..5); meanwhile the operands of pairs 0-2: row r of column 0 | row r of column 1
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          BM& b = ((3 + dy) & 1) ? bm1 : bm0;
          BM& bn = ((3 + dy) & 1) ? bm0 : bm1;
          if (dy < 2) load_bm(bn, (dy + 1) * 3 + 1); else load_bx(bx0, 0);
#pragma unroll
          for (int r = 2 * dy; r < 2 * dy + 2 && r < R; ++r) xa[r] = load_x(0, r, 1, r);
          __builtin_amdgcn_sched_barrier(0);
          main_step(b, h1, dy, false);
          __builtin_amdgcn_sched_barrier(0);
        }
        // pairs 0-2: taps (dy, 0) + (dy, 1); meanwhile the h planes of column 2
#pragma unroll
        for (int pr = 0; pr < 3; ++pr) {
          BX& b = (pr & 1) ? bx1 : bx0;
          BX& bn = (pr & 1) ? bx0 : bx1;
          if (pr < 2) load_bx(bn, pr + 1); else load_bm(bm0, 2);
          load_h(h0, 2, 2 * pr, 2 * pr + 2);
          __builtin_amdgcn_sched_barrier(0);
          cross_step(b, xa, pr);
          __builtin_amdgcn_sched_barrier(0);
        }
        // column 2 (steps 6..8); meanwhile the operands of pairs 3-4: row r | row r + 1 of column 2 (the last one repeats
        // its row: that slot meets zero weights, but must hold finite numbers)
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          BM& b = (dy & 1) ? bm1 : bm0;
          BM& bn = (dy & 1) ? bm0 : bm1;
          if (dy < 2) load_bm(bn, (dy + 1) * 3 + 2); else load_bx(bx1, 3);
#pragma unroll
          for (int r = 2 * dy; r < 2 * dy + 2 && r < R; ++r) xa[r] = load_x(2, r, 2, r + 1 < R ? r + 1 : r);
          __builtin_amdgcn_sched_barrier(0);
          main_step(b, h0, dy, false);
          __builtin_amdgcn_sched_barrier(0);
        }
        // The ninth tap (2, 2) takes bytes 16-31 of a scaled MFMA whose bytes 0-15 belong to the ninth tap of the PREVIOUS
        // chunk (x9, kept in registers).  a.pair9 (every layer whose workgroups take an even number of chunks): even chunks
        // skip the instruction, odd chunks run it for both -- 9 cross MFMAs per two chunks instead of 10; their pair-4 weights
        // hold both taps (weight_pack_x8_kernel).  Without pair9 every chunk runs it against zero weights in bytes 0-15.
        load_bx(bx0, 4);
        int4v cur[MW];
#pragma unroll
        for (int m = 0; m < MW; ++m) cur[m] = *(const int4v*)(halo + a_off[m + 2][2] + KG * PPP * 16);
        __builtin_amdgcn_sched_barrier(0);
        cross_step(bx1, xa, 0);      // pair 3: taps (0, 2) + (1, 2)
        __builtin_amdgcn_sched_barrier(0);
        if (!(a.pair9 && ((c - c_begin) & 1) == 0)) {
          int8v xq[R];
#pragma unroll
          for (int m = 0; m < MW; ++m) xq[m] = __builtin_shufflevector(x9[m], cur[m], 0, 1, 2, 3, 4, 5, 6, 7);
          cross_step(bx0, xq, 0);
        }
#pragma unroll
        for (int m = 0; m < MW; ++m) x9[m] = cur[m];
        __builtin_amdgcn_sched_barrier(0);
        WS_STAMP(1)
        continue;
      }
      AFrag fa0, fa1;
      BFrag fb0, fb1;
      load_a(fa0, halo, 0);
      load_b(fb0, slab, 0);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        AFrag& fa = (dx & 1) ? fa1 : fa0;
        AFrag& fan = (dx & 1) ? fa0 : fa1;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int step = dx * 3 + dy;
          BFrag& fb = (step & 1) ? fb1 : fb0;
          BFrag& fbn = (step & 1) ? fb0 : fb1;
          if (step + 1 < 9) {
            const int ndx = (step + 1) / 3, ndy = (step + 1) % 3;
            load_b(fbn, slab, ndy * 3 + ndx);
          }
          if (dx < 2) load_a_part(fan, halo, dx + 1, 2 * dy, 2 * dy + 2);
          __builtin_amdgcn_sched_barrier(0);
#ifdef UNETPP_WS_DBG
          if (!(a.dbg & 4))
#endif
          {
            if (step == 0 && c == c_begin && from_zero) run_mfma_first(fa, fb);
            else run_mfma(fa, fb, dy);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      WS_STAMP(1)
    }
    // ---- epilogue from registers, two row pairs; the producers are already filling the next tile's first chunk
#ifdef UNETPP_WS_DBG
    if (a.dbg & 256) { if (acc[0][0][0] == 12345.f && acc[MW - 1][NW - 1][5] == 7.f) a.status[1] = 1; continue; }
#endif
    // ---- split-K (a.ksplit > 1; small batches, where a layer has fewer tiles than the chip has CUs): this wave's raw
    // accumulators go to its slot of the partial buffer; a counter per (tile, wave) tells the wave whose partial arrives
    // LAST, and that wave adds all KS partials in the fixed order 0 .. KS-1 (its own included, read back like the others:
    // the same bits whoever is last) and runs the epilogue.  No wave ever waits for another workgroup.
    // Visibility between workgroups (per-XCD L2s, per-CU L1s: MI355X_MICROARCH.md, valid hand-off forms): every partial is
    // stored write-through (sc1) and drained (vmcnt 0) by the wave that then adds to the counter (agent-scope atomic); the
    // last wave reads every partial with sc1 loads.  No release / acquire fence (an agent-scope release writes back the whole
    // L2: measured +30 us per split launch).
    if (KS > 1) {
      typedef __attribute__((ext_vector_type(4))) float f32x4;
      const int tile_id = tile / KS;                                     // (n, ty, tx, ct): the same for the KS workgroups of a tile
      const int ntile = total_tiles / KS;
      constexpr int WAVE_BYTES = MW * NW * 16 * 64 * 4;
      const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc((void*)a.kpart, 0, KS * ntile * C::NCONS * WAVE_BYTES, 0x00020000);
      const int mine = ((ks * ntile + tile_id) * C::NCONS + cw) * WAVE_BYTES + lane * 16;
#pragma unroll
      for (int m = 0; m < MW; ++m)
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            // (whole-vector bit casts only: __builtin_bit_cast of a vector ELEMENT reads the vector's first element whatever the
            // index -- hipcc, ROCm 7.2)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, (f32x4){acc[m][j][4 * q], acc[m][j][4 * q + 1], acc[m][j][4 * q + 2], acc[m][j][4 * q + 3]}),
                                                   prs, mine + ((m * NW + j) * 4 + q) * 1024, 0, 16 /* sc1 */);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned old = 0;
      if (lane == 0) old = __hip_atomic_fetch_add(a.kcnt + tile_id * C::NCONS + cw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      old = __builtin_amdgcn_readfirstlane(old);
      if (old != (unsigned)(KS - 1)) { WS_STAMP(2) continue; }         // another workgroup's wave finishes this part of the tile
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");           // (compiler only: the loads stay behind the add)
      if (lane == 0) __hip_atomic_store(a.kcnt + tile_id * C::NCONS + cw, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      // Units of half a partial (8 of its 16 KB-pieces, 32 registers), two units in flight; every accumulator element still
      // receives its partials in the order 0, 1, 2, ...
      constexpr int HALF = MW * NW * 2;
      auto fetch = [&](f32x4 (&pv)[HALF], int u) {
        const int src = (((u >> 1) * ntile + tile_id) * C::NCONS + cw) * WAVE_BYTES + lane * 16 + (u & 1) * HALF * 1024;
#pragma unroll
        for (int t = 0; t < HALF; ++t) pv[t] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, src + t * 1024, 0, 16 /* sc1 */));
      };
      auto add = [&](const f32x4 (&pv)[HALF], int u) {
#pragma unroll
        for (int m = 0; m < MW; ++m)
#pragma unroll
          for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int t = (m * NW + j) * 4 + q;
              if ((t >= HALF) != ((u & 1) != 0)) continue;            // the other half of the partial
#pragma unroll
              for (int i = 0; i < 4; ++i) acc[m][j][4 * q + i] = u < 2 ? pv[t % HALF][i] : acc[m][j][4 * q + i] + pv[t % HALF][i];
            }
      };
      f32x4 pa[HALF], pb[HALF];
      fetch(pa, 0);
      for (int u = 0; u < 2 * KS; u += 2) {
        fetch(pb, u + 1);
        add(pa, u);
        if (u + 2 < 2 * KS) fetch(pa, u + 2);
        add(pb, u + 1);
      }
    }
#pragma unroll
    for (int m2 = 0; m2 < MW / 2; ++m2)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const bool give = handoff && (MW == 4 ? m2 == 1 : j == 1);      // this pair goes to producer wave cw
        if (give) {
          char* st = smem + (2 * C::BUF_BYTES + 2 * C::LS_BYTES + 2 * C::PATCH_BYTES) + cw * C::STG_WAVE + lane * 16;
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              typedef __attribute__((ext_vector_type(4))) float f32x4;
              const float16v& v = acc[2 * m2 + i][j];
              *(f32x4*)(st + (i * 4 + q) * 1024) = (f32x4){v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            }
        } else {
          const float16v pair[2] = {acc[2 * m2][j], acc[2 * m2 + 1][j]};
          ws_epilogue<P, 2, POOL, HEAD, X8>(a, pair, sb_lds, head_lds, n, y0 + cw * MW + 2 * m2, x0, ct * BN + j * 32, lane);
        }
      }
    WS_STAMP(2)
  }
  if (handoff) lds_barrier();        // the last handed-over pair is in LDS
#ifdef UNETPP_WS_DBG
  if (a.stamps && cw == 0 && lane == 0)
#pragma unroll
    for (int i = 0; i < 4; ++i) a.stamps[((size_t)blockIdx.x * 2) * 16 + i] = st_sum[i];
  if (a.stamps && cw == 0 && lane == 0) {
    a.stamps[((size_t)blockIdx.x * 2) * 16 + 4] = rt_begin;
    a.stamps[((size_t)blockIdx.x * 2) * 16 + 6] = rt_entry;
    a.stamps[((size_t)blockIdx.x * 2) * 16 + 5] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}
#undef WS_STAMP

}  // namespace unetpp
