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
//   * the input halo  (TH+2) x 34 pixels x KC channels — buffer-addressed LDS-DMA (buffer_load_dwordx4 ... lds)
//     straight from HBM/L2 while the previous chunk computes; pixels outside the image carry an out-of-range
//     buffer offset and arrive as zeros (the convolution's padding), no registers or ds_write in between;
//   * the weight slab 9 x KC x BN — pre-packed in exactly the order the B fragments are read, so it is
//     a linear copy, also by LDS-DMA (global_load_lds_dwordx4), in flight during the MFMAs.
// Activation layout in HBM: channel-blocked NHWC, [N][C/16][H][W][P][16] fp16 (block of CB = min(16, C)
// channels; P = 1 (FAST) or 2 (EXACT: plane 0 = hi, plane 1 = lo, value = hi + lo)).  One K-chunk of 16
// channels is therefore one contiguous 32*P bytes per pixel and consecutive pixels are contiguous: a halo
// row is one 34 x 32*P-byte run and every fetched 128-byte line is fully used (with plain [pixel][C] each
// chunk touched 32 of every 64+ bytes and the level-0/1 layers over-read HBM 2-3x: profiles/README.md).
// EXACT issues three MFMAs per product (lo*hi, hi*lo, hi*hi) into one fp32 accumulator.
//
// LDS images (bytes):
//   halo   1 KiB pieces (one DMA wave-instruction each) of [unit][PPP pixels][8 halves], unit = plane*KG +
//          k-group, PPP = 64 / (P*KG) consecutive halo pixels; the A-fragment ds_read_b128 of a 32x16 tile
//          reads contiguous runs of PPP*16 bytes -> conflict-free.
//   slab   [P][tap][KG][BN][8 halves]           B-fragment read = 2 x 512 contiguous bytes.
// Workgroups are persistent: each walks tiles blockIdx.x, +gridDim.x, ... and prefetches the first chunk
// of its next tile during the last chunk of the current one, so only the first tile pays the cold
// HBM burst.  The epilogue runs from registers (see run_mfma / pack_store): no LDS, no barrier.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace unetpp {

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
typedef __attribute__((ext_vector_type(16))) float float16v;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

struct ConvArgs {
  const half_t* in0;     // [N][C0/CB][H][W][P][CB], CB = min(16, C0)
  const half_t* in1;     // same layout with C1 channels, or nullptr (then C1 = 0)
  const half_t* wpk;     // packed weights [ct][chunk][P][tap][KG][BN][8]
  const float* scale;    // [Cout] 2^-k undoing the per-channel weight scaling
  const float* bias;     // [Cout] folded conv+BN bias
  half_t* out;           // [N][Cout/16][H][W][P][16]
  half_t* pool_out;      // [N][Cout/16][H/2][W/2][P][16] or nullptr
  int N, H, W, C0, C1, Cout;
  int tiles_x, tiles_y;  // spatial tiles per image
  int nct;               // Cout / BN
  int nchunks;           // ceil(C0 / KC) + C1 / KC
  // split-K of the wave-specialised kernel (conv3x3_ws.h): ksplit workgroups per tile, raw fp32 partials, arrival counters
  int ksplit;            // >= 1, divides nchunks
  float* kpart;          // [ksplit][tiles][4 consumer waves][MW * NW * 16 * 64] fp32
  unsigned* kcnt;        // [tiles][4], zero between launches
  int gdec[5];           // wave-specialised kernel: the grid size in the tile-number radix (ks, ct, tx, ty, n), by the host (launch_ws_k)
  int pair9;             // EXACT8: the ninth tap of an even chunk shares a scaled MFMA with the next chunk's (conv3x3_ws.h)
  // fused 1x1 head + argmax (HEAD variant, Cout == 32 == BN): self.final (unetpp.py:85,119) and the
  // frame-loop tail softmax->argmax->uint8, (pred==1), (pred==2) (infer_two_stage_burr.py:299-304)
  const float* head_w;   // [C][32] fp32
  const float* head_b;   // [C]
  int head_C;
  float* logits;         // [N][C][H][W] fp32 or nullptr
  uint8_t* mask;         // [N][H][W] or nullptr
  uint8_t* cable;        // or nullptr
  uint8_t* tape;         // or nullptr
  // probability outputs: softmax over the C logits in fp32 and the class rules the
  // thresholded frame loops apply to it; rule 0 = plain (pred==1)/(pred==2) of infer_two_stage_burr.py
  float* probs;          // [N][C][H][W] fp32 or nullptr
  int rule;              // UNETPP_RULE_*
  float t_cable, t_tape, bg_margin, ct_margin;
  unsigned* status;      // engine's sticky range flags (ST_*), see range_flag
  // fused first ConvBlock (conv3x3_ws.h, C0F): the caller's input tensor and conv0_0.conv1's packed weights
  const float* zinit;    // ZINIT: [N][Cout/32][H*W][32] fp32 accumulator start values (tapmm_ws.h), else unused
  const void* raw_in;    // float32 [N,3,H,W] (raw_fmt 0) or uint8 [N,H,W,3] BGR (raw_fmt 1)
  int raw_fmt;
  const half_t* c1w;     // [half 2][plane 2][lane 64][8]: A fragments of v_mfma_f32_16x16x32_f16, see conv0_pack_kernel
  const float* c1_scale; // [32]
  const float* c1_bias;  // [32]
#ifdef UNETPP_WS_DBG
  int dbg;               // measurement builds only: phases of conv3x3_ws_kernel switched off (results are garbage)
  unsigned long long* stamps;   // or nullptr: [workgroup][role][8] cycle sums of the phases of one wave per role
#endif
};

// ---- range status -------------------------------------------------------------------------------------
// Activations are stored as fp16 hi + lo, so a value beyond +-65504 cannot be represented: it is clamped, and a
// NaN does not survive the ReLU (v_max_f32 returns the other operand).  The fp32 reference
// (src/models/unetpp.py:23-26, simple_unet.py:94-128) has neither limit, so every kernel that narrows a value
// reports it in the engine's sticky status word (include/unetpp.h: unetpp_status).  NaNs can only enter through
// the input (convert_input_kernel checks every value) or through non-finite weights (weight_scale_kernel /
// convt_scale_kernel check them at load time): finite fp16 operands cannot overflow the fp32 accumulator.  The
// conv epilogues therefore only watch the fp16 ceiling: half a VALU instruction per value (v_max3_f32).
constexpr unsigned ST_OVERFLOW = 1u, ST_NAN = 2u;
constexpr float F16_MAX = 65504.0f;
__device__ __forceinline__ void range_flag(unsigned* status, bool out_of_range, bool is_nan) {
  const unsigned long long bn = __builtin_amdgcn_ballot_w64(is_nan);
  const unsigned long long bo = __builtin_amdgcn_ballot_w64(out_of_range && !is_nan);
  if ((threadIdx.x & 63) == 0 && status) atomicOr(status, (bn ? ST_NAN : 0u) | (bo ? ST_OVERFLOW : 0u));
}

// cable/tape decision for one pixel from its class probabilities (p0 = background, p1 = cable, p2 = tape)
//   1 thresholded_argmax              infer_video_3class_best.py:56-83, infer_video_strict.py:36-63
//   2 strict_threshold_with_bg_check  infer_video_fixed.py:35-83
//   3 exclusive_threshold             infer_video_robust.py:70-99
__device__ __forceinline__ void apply_rule(int rule, float p0, float p1, float p2, float tc, float tt, float bgm,
                                           float ctm, bool& cable, bool& tape) {
  const int winner = (p1 > p0) ? ((p2 > p1) ? 2 : 1) : ((p2 > p0) ? 2 : 0);   // np.argmax: first maximum
  if (rule == 1) {
    cable = winner == 1 && p1 >= tc && (p1 - p0) >= bgm;
    tape = winner == 2 && p2 >= tt && (p2 - p0) >= bgm;
  } else if (rule == 2) {
    cable = winner == 1 && p1 >= tc && p0 <= bgm;
    tape = winner == 2 && p2 >= tt && p0 <= bgm;
  } else {
    const bool cand_c = p1 >= tc && p1 >= p0 + bgm;
    const bool cand_t = p2 >= tt && p2 >= p0 + bgm;
    cable = cand_c && p1 >= p2 + ctm;
    tape = cand_t && p2 >= p1 + ctm;
    if (cable && tape) { cable = p1 >= p2; tape = !cable; }
  }
}

template <int P, int KC, int NW, int MW, int WAVES, bool SINGLE = false, bool UPF = false>
struct ConvCfg {
  static constexpr int NT = WAVES * 64;
  static constexpr int TH = WAVES * MW, TW = 32, HALO_W = TW + 2, NHALO = (TH + 2) * HALO_W;
  static constexpr int KG = KC / 8, BN = 32 * NW;
  // Halo image in LDS = a sequence of 1 KiB pieces, each written by ONE LDS-DMA wave-instruction:
  //   piece = PPP consecutive halo pixels x U units, laid out [unit][pixel][8 halves]
  //   unit  = plane * KG + k-group (8 channels) -- U = P * KG of them, PPP = 64 / U pixels per piece
  // so a fragment read (32 consecutive pixels of one unit) is made of contiguous runs of PPP * 16 bytes
  // (a multiple of 256 B: conflict-free ds_read_b128, also where a run wraps into the next piece).
  static constexpr int U = P * KG;
  static constexpr int PPP = 64 / U;
  static constexpr int HALO_PIECES = (NHALO + PPP - 1) / PPP;
  static constexpr int HALO_BYTES = HALO_PIECES * 1024;
  static constexpr int HALO_ITERS = (HALO_PIECES + WAVES - 1) / WAVES;   // DMA pieces per wave and chunk
  static constexpr int SLAB_BYTES = P * 9 * KC * BN * 2;
  static constexpr int BUF_BYTES = HALO_BYTES + SLAB_BYTES;
  // Stage buffers per workgroup: two (the next chunk loads under this chunk's MFMAs; one workgroup per CU in exact
  // mode), or with SINGLE one, so that two workgroups fit a CU.  The load, matrix and epilogue phases of a single
  // workgroup's lock-stepped waves do not overlap each other (measured: they add up); those of two workgroups do.
  // It pays where HBM is not the limit anyway: the fused-head conv, which reads x0_4a and writes one byte per pixel
  // (265 -> 221 us); the other full-resolution convs already run at the HBM rate and stay double-buffered.
  static constexpr int STAGES = SINGLE ? 1 : 2;
  // UPF (fused bilinear upsample, see the kernel): the low-res pixels one chunk of `up` channels is interpolated
  // from -- at most TH/2+2 rows x 18 columns (checked for every tile origin and size up to 4096) of P*32-byte
  // records -- staged by LDS-DMA two chunks ahead, double buffered.
  static constexpr int LSH = TH / 2 + 2, LSW = TW / 2 + 2, LS_PX = LSH * LSW;
  static constexpr int LS_REC = P * 32;                                   // bytes per low-res pixel and channel block
  static constexpr int LS_PIECES = (LS_PX * LS_REC + 1023) / 1024;
  static constexpr int LS_BYTES = UPF ? LS_PIECES * 1024 : 0;
  static constexpr int LS_ITERS = (LS_PIECES + WAVES - 1) / WAVES;
  static constexpr int LDS_BYTES = STAGES * BUF_BYTES + 2 * LS_BYTES;
  static constexpr int SLAB_PIECES = SLAB_BYTES / 1024;   // one LDS-DMA wave-instruction = 1 KiB
  static_assert(U == 2 || U == 4, "unit count per pixel");
  static_assert(SLAB_BYTES % 1024 == 0, "slab must be a whole number of 1 KiB DMA pieces");
  static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ void split_f16(float v, half_t& hi, half_t& lo) {
  v = fminf(v, F16_MAX);
  hi = (half_t)v;
  lo = (half_t)(v - (float)hi);
}

typedef __attribute__((address_space(3))) char lds_char_t;

// LDS-DMA of 64 lanes x 16 B: LDS destination = wave-uniform byte address `lds_dst` + lane*16, global
// source per lane.  Issued from inline asm on purpose: hipcc (ROCm 7.2) treats the builtin form as a
// flat access that may alias LDS and then degrades every later `s_waitcnt lgkmcnt(N)` of the MFMA loop
// to lgkmcnt(0) (no ds_read prefetch overlap).  The asm is invisible to the compiler's counters, so
// the consumer side waits explicitly: `s_waitcnt vmcnt(0)` before the barrier that publishes the buffer.
// A kernel's argument block spans several cache lines and the compiler fetches a field when it first needs it: every first
// touch of a line is a miss of the scalar cache (invalidated at the launch boundary), one after the other on the way to the
// first DMA -- about a microsecond per launch with ConvArgs' nine lines (profiles/r03_b1_timeline.txt).  Touch all lines in
// one burst at kernel entry; the fields hit afterwards.
template <class Args>
__device__ __forceinline__ void touch_kernarg_lines() {
  const int* ka = (const int*)__builtin_amdgcn_kernarg_segment_ptr();
  int touched = 0;
#pragma unroll
  for (int o = 0; o < (int)(sizeof(Args) / 4); o += 16) touched |= ka[o];
  asm volatile("" :: "s"(touched));
}

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_dst)
               : "memory");
}

// The same for a buffer-addressed source: per-lane byte offset `voff` into the buffer `rsrc` plus the
// wave-uniform `soff`; lanes whose offset lies outside the buffer write ZEROS to their 16 bytes of LDS
// (checked on gfx950) -- which is how the convolution's zero padding gets into the halo image.
template <bool NT = false>
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_dst) {
  unsigned keep;
  // NT: the `nt` cache policy for bytes this launch reads (almost) once -- a conv's halo rows: one frame per call takes 4-5 % less (the first chunk
  // lands sooner), batch 16 is unchanged (profiles/r03_halo_nt_ab.txt).  Not for tiles that neighbours re-read
  // out of the L2 (the low-resolution GEMM's operand, the up-sum's Y): those got slower.
  if (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst)
                 : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst)
                 : "memory");
}

// Epilogue helper shared by the conv and transposed-conv kernels.  `v` = the 16 channel values of one pixel
// held by this lane (channel (r & 3) + 8 * (r >> 2) + 4 * h of a 32-channel tile, h = lane >> 5).  Packs to
// fp16 (hi/lo planes when P == 2), exchanges words with lane ^ 32 (v_permlane32_swap) so that every lane
// owns 8 consecutive channels, and stores 16 bytes per plane and channel block.
// dst -> plane 0 of the pixel in the tile's first channel block; blk_stride = halves between channel blocks.
// (hi, lo) fp16 planes of two fp32 values, packed: hi = RNE(v), lo = RNE(v - hi).  v - hi is exact in fp32, so the
// mixed-precision FMA (fp16 operand read straight out of the packed word) gives the same bits as cvt + sub with
// 4 instead of 10 instructions per pair.  The caller guarantees |v| <= 65504 (see range_flag).
__device__ __forceinline__ void split_pack2(float v0, float v1, unsigned& wh, unsigned& wl) {
  half2v ph = {(half_t)v0, (half_t)v1};                 // v_cvt_pk_f16_f32
  wh = __builtin_bit_cast(unsigned, ph);
  // lo = fp16(v - hi): one mixed-precision FMA per value reads hi out of the packed word, subtracts in fp32 and writes
  // the rounded result straight into its half of the lo word (no separate conversion / pack)
  unsigned l;
  asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(wh), "v"(v0));
  asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(wh), "v"(v1));
  wl = l;
}

// ---- EXACT8 (conv3x3_ws.h): an activation is stored as fp16 hi (RNE) plus two e5m2 bytes per value,
//   lo8 = e5m2(2^8 (v - hi))   the residual, as the 8-bit operand of the  lo * w  cross term
//   x8  = e5m2(2^-3 v)         the value itself, as the 8-bit operand of the  x * w_lo  cross term
// (pre-scaled so that neither can overflow or go subnormal where it matters: |v - hi| <= 32, v <= 65504; the scaled
// MFMA undoes both factors with one block scale).  A 64-byte pixel record of 16 channels is
//   [hi c0-7][hi c8-15][lo8 c0-3, x8 c0-3, lo8 c4-7, x8 c4-7][the same for c8-15]
// so that the lanes of every kernel write and read the same 16-byte pieces as in the two-plane fp16 format.
typedef __attribute__((ext_vector_type(2))) short short2v;
constexpr float X8_LO_DIV = 0.00390625f, X8_X_DIV = 8.0f;      // v_cvt_scalef32_pk_bf8_f32 divides by its scale operand ...
constexpr float X8_LO_MUL = 0.00390625f;                       // ... v_cvt_scalef32_f32_bf8 multiplies by it (scripts/microbench/bf8_cvt_probe.hip)
// the E8M0 block scales of v_mfma_scale_f32_32x32x64_f8f6f4 for (weights e4m3, activations e5m2): 2^6 and 2^-8, see weight_pack_x8_kernel
constexpr int X8_SCALE_W = 127 + 6, X8_SCALE_A = 127 - 8;
// four channel values -> two packed fp16 hi words, one word of four lo8 bytes, one word of four x8 bytes
__device__ __forceinline__ void split_pack4_x8(float v0, float v1, float v2, float v3, unsigned& h0, unsigned& h1, unsigned& l8, unsigned& x8) {
  half2v p0 = {(half_t)v0, (half_t)v1}, p1 = {(half_t)v2, (half_t)v3};      // v_cvt_pk_f16_f32
  h0 = __builtin_bit_cast(unsigned, p0); h1 = __builtin_bit_cast(unsigned, p1);
  float l0, l1, l2, l3;                                                       // v - hi, exact in fp32
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l0) : "v"(h0), "v"(v0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l1) : "v"(h0), "v"(v1));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l2) : "v"(h1), "v"(v2));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(l3) : "v"(h1), "v"(v3));
  short2v t = {0, 0};
  t = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t, l0, l1, X8_LO_DIV, false);
  t = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(t, l2, l3, X8_LO_DIV, true);
  l8 = __builtin_bit_cast(unsigned, t);
  short2v u = {0, 0};
  u = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(u, v0, v1, X8_X_DIV, false);
  u = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(u, v2, v3, X8_X_DIV, true);
  x8 = __builtin_bit_cast(unsigned, u);
}

template <int P, bool X8 = false>
__device__ __forceinline__ void pack_store_octets(const float (&v)[16], half_t* dst, size_t blk_stride, bool ok, int h) {
  unsigned wh[4][2], wl[4][2];
  if (X8) {      // EXACT8 records: wl[q] = {lo8 x 4, x8 x 4} of the quad's four channels; the same exchange then builds the octet's 16 bytes
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pack4_x8(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3], wh[q][0], wh[q][1], wl[q][0], wl[q][1]);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2) {
      const float v0 = v[4 * q + 2 * w2], v1 = v[4 * q + 2 * w2 + 1];      // within the fp16 range: the caller clamped
      if (X8) {
      } else if (P == 2) {
        split_pack2(v0, v1, wh[q][w2], wl[q][w2]);
      } else {
        half2v ph = {(half_t)v0, (half_t)v1};
        wh[q][w2] = __builtin_bit_cast(unsigned, ph);
      }
    }
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
#pragma unroll
    for (int w2 = 0; w2 < 2; ++w2) {
      auto r = __builtin_amdgcn_permlane32_swap(wh[2 * pi][w2], wh[2 * pi + 1][w2], false, false);
      wh[2 * pi][w2] = r[0]; wh[2 * pi + 1][w2] = r[1];
      if (P == 2) {
        auto r2 = __builtin_amdgcn_permlane32_swap(wl[2 * pi][w2], wl[2 * pi + 1][w2], false, false);
        wl[2 * pi][w2] = r2[0]; wl[2 * pi + 1][w2] = r2[1];
      }
    }
    if (ok) {      // octet pi*16 + 8h of this 32-channel tile = channel block pi, halves 8h..8h+7
      u32x4 o = {wh[2 * pi][0], wh[2 * pi][1], wh[2 * pi + 1][0], wh[2 * pi + 1][1]};
      *(u32x4*)(dst + pi * blk_stride + 8 * h) = o;
      if (P == 2) {
        u32x4 o2 = {wl[2 * pi][0], wl[2 * pi][1], wl[2 * pi + 1][0], wl[2 * pi + 1][1]};
        *(u32x4*)(dst + pi * blk_stride + 16 + 8 * h) = o2;
      }
    }
  }
}

// The same for accumulators of weights packed with rows8 (weight_pack_kernel): registers 0..7 are channels 8 h .. 8 h + 7
// of the tile's first record, registers 8..15 the same eight of the second -- the stores of pack_store_octets without
// its exchange between the half-waves.
template <int P, bool X8 = false>
__device__ __forceinline__ void pack_store_rows8(const float (&v)[16], half_t* dst, size_t blk_stride, bool ok, int h, bool dbg_nostore = false, bool dbg_coalesced = false) {
#pragma unroll
  for (int pi = 0; pi < 2; ++pi) {
    unsigned wh[4], wl[4];
    if (X8) {
      split_pack4_x8(v[8 * pi], v[8 * pi + 1], v[8 * pi + 2], v[8 * pi + 3], wh[0], wh[1], wl[0], wl[1]);
      split_pack4_x8(v[8 * pi + 4], v[8 * pi + 5], v[8 * pi + 6], v[8 * pi + 7], wh[2], wh[3], wl[2], wl[3]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (X8) {
      } else if (P == 2) {
        split_pack2(v[8 * pi + 2 * i], v[8 * pi + 2 * i + 1], wh[i], wl[i]);
      } else {
        half2v ph = {(half_t)v[8 * pi + 2 * i], (half_t)v[8 * pi + 2 * i + 1]};
        wh[i] = __builtin_bit_cast(unsigned, ph);
      }
    }
#ifdef UNETPP_WS_DBG
    if (dbg_nostore) {      // timing experiment: the whole epilogue except its global stores
      asm volatile("" :: "v"(wh[0]), "v"(wh[1]), "v"(wh[2]), "v"(wh[3]));
      if (P == 2) asm volatile("" :: "v"(wl[0]), "v"(wl[1]), "v"(wl[2]), "v"(wl[3]));
      continue;
    }
    if (dbg_coalesced) {    // timing experiment: the same bytes to the same 2 KB row, one contiguous KB per instruction (WRONG data placement)
      const int lane_ = (int)(threadIdx.x & 63), p_ = lane_ & 31;
      half_t* row = dst - p_ * (P * 16) + pi * blk_stride;
      if (ok) {
        *(u32x4*)(row + lane_ * 8) = (u32x4){wh[0], wh[1], wh[2], wh[3]};
        if (P == 2) *(u32x4*)(row + 512 + lane_ * 8) = (u32x4){wl[0], wl[1], wl[2], wl[3]};
      }
      continue;
    }
#endif
    if (ok) {
      // (streaming `nt` stores here: one frame per call 2 % sooner in the layer table, nothing in bench.py's median, batch 16 0.6-1 %
      // later -- also as a per-launch switch for small launches; not kept.  profiles/r03_halo_nt_ab.txt)
      *(u32x4*)(dst + pi * blk_stride + 8 * h) = (u32x4){wh[0], wh[1], wh[2], wh[3]};
      if (P == 2) *(u32x4*)(dst + pi * blk_stride + 16 + 8 * h) = (u32x4){wl[0], wl[1], wl[2], wl[3]};
    }
  }
}

constexpr int HEAD_MAX_CLASSES = 16;
constexpr int HEAD_FUSED_MAX_CLASSES = 8;   // the fused head keeps all logits in registers

// the exact-mode fused-head kernel runs single-staged, two workgroups per CU (see ConvCfg::STAGES)
template <int P, bool HEAD> constexpr bool conv_single_stage() { return HEAD && P == 2; }

// UPF = fused bilinear upsample (reference unetpp.py:76,112-116: cat([skip, self.up(low)])): the second source `in1`
// is then the LOW-resolution tensor [N][C1/16][H/2][W/2][P][16] itself; the loader interpolates each chunk's halo
// image from it (align_corners=True: src = dst*(in-1)/(out-1), same arithmetic as upsample2x_kernel) instead of
// fetching a materialised `up` tensor: per tile and chunk, <= 10 x 18 low-res pixel records arrive by LDS-DMA two
// chunks ahead, and while chunk c multiplies, the waves build chunk c+1's halo image from them (VALU + ds_write
// beside the MFMAs).  The `up` tensor is never written or read: -2.4 GB of the 12.7 GB step at level 0.
// ZINIT: the accumulators of a tile start from a.zinit (the low-resolution half of a decoder conv, tapmm_ws.h)
// instead of zero; K then runs over the skip channels only.
template <int P, int KC, int NW, int MW, int WAVES, bool POOL, bool HEAD = false, bool UPF = false, bool ZINIT = false>
__global__ __launch_bounds__(WAVES * 64, (conv_single_stage<P, HEAD>() ? 2 : 1))
void conv3x3_bias_relu_kernel(ConvArgs a) {
  touch_kernarg_lines<ConvArgs>();
  using C = ConvCfg<P, KC, NW, MW, WAVES, conv_single_stage<P, HEAD>(), UPF>;
  static_assert(!UPF || (!POOL && !HEAD && KC == 16 && MW == 2 && !conv_single_stage<P, HEAD>()), "fused upsample: plain 16-row tiles only");
  constexpr int NT = C::NT, TH = C::TH, TW = C::TW, HALO_W = C::HALO_W;
  constexpr int KG = C::KG, BN = C::BN, PPP = C::PPP;
  constexpr int ITERS = C::HALO_ITERS;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = a.H, W = a.W;
  const int tiles_img = a.tiles_x * a.tiles_y;
  const int total_tiles = a.N * tiles_img * a.nct;
  const int nch0 = (a.C0 + KC - 1) / KC;
  const unsigned lds_base = (unsigned)(unsigned long)(lds_char_t*)smem;

  // ---- tile-invariant halo geometry: wave w issues the DMA pieces w, w + WAVES, ...; in a piece, lane l
  // fetches unit l / PPP of halo pixel piece * PPP + l % PPP
  const int my_u = lane / PPP;
  const int my_pl = my_u / KG, my_k8 = (my_u % KG) * 8;    // plane and first channel (within the chunk) of this lane's unit
  int hyx[ITERS];      // (hy << 8) | hx of the halo pixel, -1 = no pixel (tail of the last piece)
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int hp = (wave + it * WAVES) * PPP + lane % PPP;
    const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
    hyx[it] = (hp < C::NHALO) ? ((hy << 8) | hx) : -1;
  }

  // ---- per-tile state: the tile being computed (cur_*) and the source state of the tile whose
  // chunks are being loaded (voff*, rsrc*, wsrc) — the latter switches to the next tile one chunk early.
  constexpr unsigned OOB = 0x80000000u;   // beyond num_records: the buffer load returns zeros
  unsigned voff0[ITERS], voff1[ITERS];
  __amdgpu_buffer_rsrc_t rsrc0, rsrc1;
  const char* wsrc;
  const unsigned img_bytes0 = (unsigned)(H * W * P * a.C0 * 2);
  const unsigned img_bytes1 = (unsigned)((UPF ? (H >> 1) * (W >> 1) : H * W) * P * a.C1 * 2);
  int cur_n, cur_y0, cur_x0, cur_ct;
  auto decode = [&](int t, int& n, int& y0, int& x0, int& ct) {
    ct = t % a.nct;
    int pt = t / a.nct;
    const int tx = pt % a.tiles_x; pt /= a.tiles_x;
    const int ty = pt % a.tiles_y;
    n = pt / a.tiles_y;
    x0 = tx * TW; y0 = ty * TH;
  };
  // Raw buffer loads: one descriptor per source covering image n, per-lane byte offset computed once per
  // tile, the chunk's channel offset in the scalar soffset; pixels outside the image (zero padding) and
  // unused items carry an out-of-range offset and read back zeros — no branches in the K loop.
  const int cb0 = a.C0 < 16 ? a.C0 : 16;                        // channel block of source 0 (8 only for the input tensor)
  const unsigned plane_bytes0 = (unsigned)(H * W * P * cb0 * 2);   // one channel block of one image
  const int Hs = UPF ? (H >> 1) : H, Ws = UPF ? (W >> 1) : W;      // extent of source 1 (UPF: the low-res tensor)
  const unsigned plane_bytes1 = (unsigned)(Hs * Ws * P * 16 * 2);
  // ---- UPF state.  Interpolation items: one (halo pixel, channel octet) per thread and round; the halo pixel of an
  // item never changes, its four low-res corners and weights are set up per tile.
  constexpr int UP_ITEMS = C::NHALO * KG;
  constexpr int UP_ROUNDS = UPF ? (UP_ITEMS + NT - 1) / NT : 0;
  constexpr int LSR = UPF ? C::LS_ITERS : 1, UPR = UPF ? UP_ROUNDS : 1;
  unsigned voffL[LSR];                       // LDS-DMA source offsets of this lane's low-res staging pieces
  int up_o00[UPR], up_o01[UPR], up_o10[UPR]; // byte offsets of the corners (y0,x0), (y0,x1), (y1,x0) in the staging image
  float up_lx1[UPR], up_ly0[UPR], up_ly1[UPR];
  int up_dst[UPR];                           // byte offset of the item's hi octet in a halo image, -1 = no item
  const float up_sh = Hs > 1 ? (float)(Hs - 1) / (float)(H - 1) : 0.f;
  const float up_sw = Ws > 1 ? (float)(Ws - 1) / (float)(W - 1) : 0.f;
  if (UPF) {
#pragma unroll
    for (int r = 0; r < UPR; ++r) {
      const int i = tid + r * NT;
      const int kg = i / C::NHALO, hp = i - kg * C::NHALO;
      up_dst[r] = i < UP_ITEMS ? (hp / PPP) * 1024 + (kg * PPP + hp % PPP) * 16 : -1;
    }
  }
  auto setup_sources = [&](int n, int y0, int x0, int ct) {
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
      const int hy = hyx[it] >> 8, hx = hyx[it] & 255;
      const int gy = y0 + hy - 1, gx = x0 + hx - 1;
      const int pl = my_pl, k8 = my_k8;                        // k8 = 8 * (k-group inside the chunk)
      const bool ok = hyx[it] >= 0 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const unsigned pix = (unsigned)((gy * W + gx) * P + pl);
      // k-groups 0,1 of a chunk sit in its first channel block, 2,3 (KC = 32) in the next one
      voff0[it] = (ok && k8 < a.C0) ? (unsigned)(k8 >> 4) * plane_bytes0 + (pix * cb0 + (k8 & 15)) * 2u : OOB;
      voff1[it] = ok ? (unsigned)(k8 >> 4) * plane_bytes1 + (pix * 16 + (k8 & 15)) * 2u : OOB;
    }
    rsrc0 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in0 + (size_t)n * H * W * P * a.C0), 0, (int)img_bytes0, 0x00020000);
    rsrc1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.in1 ? a.in1 + (size_t)n * Hs * Ws * P * a.C1 : a.in0), 0,
                                              (int)(a.in1 ? img_bytes1 : 0u), 0x00020000);
    wsrc = (const char*)a.wpk + (size_t)ct * a.nchunks * C::SLAB_BYTES;
    if (UPF) {
      // first low-res row / column any halo pixel of this tile touches (halo pixels outside the image touch none)
      const int ybase = (int)(up_sh * (float)max(y0 - 1, 0)), xbase = (int)(up_sw * (float)max(x0 - 1, 0));
      constexpr int PXP = 1024 / C::LS_REC, PARTS = C::LS_REC / 16;      // pixels per 1 KiB piece, 16-byte parts per record
#pragma unroll
      for (int it = 0; it < LSR; ++it) {
        const int lp = (wave + it * WAVES) * PXP + lane / PARTS;           // staged pixel (row-major LSH x LSW)
        const int ly = lp / C::LSW, lx = lp - ly * C::LSW;
        const int yy = min(ybase + ly, Hs - 1), xx = min(xbase + lx, Ws - 1);   // clamped: always a valid record
        voffL[it] = lp < C::LS_PX ? (unsigned)((yy * Ws + xx) * C::LS_REC + (lane % PARTS) * 16) : OOB;
      }
#pragma unroll
      for (int r = 0; r < UPR; ++r) {
        const int i = tid + r * NT;
        const int kg = i / C::NHALO, hp = i - kg * C::NHALO;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool inside = i < UP_ITEMS && gy >= 0 && gy < H && gx >= 0 && gx < W;
        // align_corners=True source coordinates, exactly as upsample2x_kernel computes them
        const float fy = up_sh * (float)max(gy, 0), fx = up_sw * (float)max(gx, 0);
        const int yy0 = min((int)fy, Hs - 1), xx0 = min((int)fx, Ws - 1);
        const int yy1 = yy0 + (yy0 < Hs - 1 ? 1 : 0), xx1 = xx0 + (xx0 < Ws - 1 ? 1 : 0);
        const float ly1 = fminf(fmaxf(fy - (float)yy0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - (float)xx0, 0.f), 1.f);
        const int ry0 = min(max(yy0 - ybase, 0), C::LSH - 1), ry1 = min(max(yy1 - ybase, 0), C::LSH - 1);
        const int rx0 = min(max(xx0 - xbase, 0), C::LSW - 1), rx1 = min(max(xx1 - xbase, 0), C::LSW - 1);
        up_o00[r] = (ry0 * C::LSW + rx0) * C::LS_REC + kg * 16;
        up_o01[r] = (ry0 * C::LSW + rx1) * C::LS_REC + kg * 16;
        up_o10[r] = (ry1 * C::LSW + rx0) * C::LS_REC + kg * 16;
        up_lx1[r] = lx1;
        up_ly0[r] = inside ? 1.f - ly1 : 0.f;           // zero padding of the convolution: both row weights 0
        up_ly1[r] = inside ? ly1 : 0.f;
      }
    }
  };

  // halo piece `it` of this wave for chunk c, straight into the halo image at LDS byte offset `halo_off`:
  // no staging registers, no ds_write.  `it` is a compile-time constant at every call site.
  auto halo_dma_one = [&](int c, int halo_off, int it) {
    const int piece = wave + it * WAVES;
    if (C::HALO_PIECES % WAVES != 0 && piece >= C::HALO_PIECES) return;
    const unsigned dst = lds_base + halo_off + piece * 1024;
    // scalar offset = first channel block of the chunk
    if (c < nch0) blds16(rsrc0, voff0[it], c * (KC / 16) * (int)plane_bytes0, dst);        // (nt measured 1.5-2 % slower in this kernel)
    else if (!UPF) blds16(rsrc1, voff1[it], (c - nch0) * (KC / 16) * (int)plane_bytes1, dst);
  };
  // UPF: staging piece `it` of this wave for up-chunk c -> staging buffer (c & 1)
  const int ls_base = C::STAGES * C::BUF_BYTES;
  auto ls_dma_one = [&](int c, int it) {
    const int piece = wave + it * WAVES;
    if (C::LS_PIECES % WAVES != 0 && piece >= C::LS_PIECES) return;
    blds16(rsrc1, voffL[it], (c - nch0) * (int)plane_bytes1, lds_base + ls_base + (c & 1) * C::LS_BYTES + piece * 1024);
  };
  // UPF: item `r` of this thread: interpolate 8 channels of one halo pixel of up-chunk c from staging buffer (c & 1)
  // into the halo image at `halo_off`.  x inside each row first, then y -- the order upsample2x_kernel uses.
  auto up_item = [&](int c, int halo_off, int r) {
    if (up_dst[r] < 0) return;
    const char* ls = smem + ls_base + (c & 1) * C::LS_BYTES;
    const int o11 = up_o10[r] + (up_o01[r] - up_o00[r]);
    const half8 h00 = *(const half8*)(ls + up_o00[r]), h01 = *(const half8*)(ls + up_o01[r]);
    const half8 h10 = *(const half8*)(ls + up_o10[r]), h11 = *(const half8*)(ls + o11);
    half8 l00, l01, l10, l11;
    if (P == 2) {
      l00 = *(const half8*)(ls + up_o00[r] + 32); l01 = *(const half8*)(ls + up_o01[r] + 32);
      l10 = *(const half8*)(ls + up_o10[r] + 32); l11 = *(const half8*)(ls + o11 + 32);
    }
    const float lx1 = up_lx1[r], lx0 = 1.f - lx1, ly0 = up_ly0[r], ly1 = up_ly1[r];
    half8 oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t0, t1;
      if (P == 2) {
        t0 = fmaf(lx1, (float)l01[e], fmaf(lx1, (float)h01[e], fmaf(lx0, (float)l00[e], lx0 * (float)h00[e])));
        t1 = fmaf(lx1, (float)l11[e], fmaf(lx1, (float)h11[e], fmaf(lx0, (float)l10[e], lx0 * (float)h10[e])));
      } else {
        t0 = fmaf(lx1, (float)h01[e], lx0 * (float)h00[e]);
        t1 = fmaf(lx1, (float)h11[e], lx0 * (float)h10[e]);
      }
      const float v = fmaf(ly1, t1, ly0 * t0);
      const half_t hi = (half_t)v;
      oh[e] = hi;
      if (P == 2) ol[e] = (half_t)(v - (float)hi);
    }
    char* dst = smem + halo_off + up_dst[r];
    *(half8*)dst = oh;
    if (P == 2) *(half8*)(dst + KG * PPP * 16) = ol;
  };
  constexpr int DMA_PER_WAVE = (C::SLAB_PIECES + WAVES - 1) / WAVES;
  auto slab_dma_one = [&](int c, int slab_off, int p) {
    const int piece = wave + p * WAVES;
    if (C::SLAB_PIECES % WAVES == 0 || piece < C::SLAB_PIECES)
      glds16(wsrc + (size_t)c * C::SLAB_BYTES + piece * 1024, lane * 16, lds_base + slab_off + piece * 1024);
  };

  // lane-constant LDS read offsets.  Pixel fragment of halo row r (0 .. MW+1 of this wave) shifted by dx:
  // lane l reads unit (l >> 5) of pixel hp = (wave*MW + r) * HALO_W + (l & 31) + dx.
  int a_off[MW + 2][3];
#pragma unroll
  for (int r = 0; r < MW + 2; ++r)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int hp = (wave * MW + r) * HALO_W + (lane & 31) + dx;
      a_off[r][dx] = (hp / PPP) * 1024 + ((lane >> 5) * PPP + hp % PPP) * 16;
    }
  const int b_lane_off = ((lane >> 5) * BN + (lane & 31)) * 16;

  struct Frag { half8 ah[MW], al[MW], bh[NW], bl[NW]; };
  auto load_frags = [&](Frag& f, const char* halo, const char* slab, int step) {
    const int tap = step % 9, s = step / 9;
    const int dy = tap / 3, dx = tap % 3;
#pragma unroll
    for (int m = 0; m < MW; ++m) {
      const char* p = halo + a_off[m + dy][dx] + (2 * s) * PPP * 16;      // k-groups 2s, 2s+1
      f.ah[m] = *(const half8*)p;
      if (P == 2) f.al[m] = *(const half8*)(p + KG * PPP * 16);             // plane 1 = units KG ..
    }
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int off = b_lane_off + ((tap * KG + 2 * s) * BN + j * 32) * 16;
      f.bh[j] = *(const half8*)(slab + off);
      if (P == 2) f.bl[j] = *(const half8*)(slab + off + 9 * KC * BN * 2);
    }
  };
  float16v acc[MW][NW];
  // D = W^T-tile x pixel-tile: the weights go in as the MFMA's A operand and the pixels as B, so the
  // accumulator has the PIXEL on the lane (col = lane & 31) and 16 output channels in its registers
  // (channel = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)): four consecutive channels sit in four
  // consecutive registers, which is what lets the epilogue pack and store 16-byte channel runs (and
  // take the 2x2 max-pool) straight from registers.
  auto run_mfma = [&](const Frag& f) {
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        if (P == 2) {
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.al[m], acc[m][j], 0, 0, 0);
          acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bl[j], f.ah[m], acc[m][j], 0, 0, 0);
        }
        acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.bh[j], f.ah[m], acc[m][j], 0, 0, 0);
      }
  };

  // per-channel (scale, bias) of the whole layer staged once per workgroup: the epilogue then reads them
  // from LDS (broadcast reads) instead of exposing a global-load latency per tile
  float2* sb_lds = (float2*)(smem + C::LDS_BYTES);
  for (int i = tid; i < a.Cout; i += NT) sb_lds[i] = make_float2(a.scale[i], a.bias[i]);
  float* head_lds = (float*)(smem + C::LDS_BYTES + a.Cout * 8);   // HEAD only: [C][32] weights then [C] biases
  if (HEAD) {
    static_assert(!HEAD || (NW == 1 && !POOL), "the fused head needs all 32 channels of x0_4 in one tile");
    for (int i = tid; i < a.head_C * 32; i += NT) head_lds[i] = a.head_w[i];
    for (int i = tid; i < a.head_C; i += NT) head_lds[a.head_C * 32 + i] = a.head_b[i];
  }

  // ---- tile order: workgroups b, b+8, b+16, ... share an XCD (round-robin dispatch), hence an L2.
  // Give each XCD a contiguous run of tiles per round, so that the nct channel tiles of one pixel
  // tile and spatially neighbouring pixel tiles (shared halo) are fetched into the same L2.
  // Speed only: any placement gives the same result.
  const int G = (int)gridDim.x;
  const int slot = (G % 8 == 0) ? ((int)blockIdx.x % 8) * (G / 8) + (int)blockIdx.x / 8 : (int)blockIdx.x;

  // ---- first tile: chunk 0 -> buffer 0
  int tile = slot;
  if (tile >= total_tiles) return;
  decode(tile, cur_n, cur_y0, cur_x0, cur_ct);
  setup_sources(cur_n, cur_y0, cur_x0, cur_ct);
#pragma unroll
  for (int it = 0; it < ITERS; ++it) halo_dma_one(0, 0, it);
#pragma unroll
  for (int p = 0; p < DMA_PER_WAVE; ++p) slab_dma_one(0, C::HALO_BYTES, p);
  constexpr int STAGES = C::STAGES;
  if (STAGES == 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  int g = 0;   // global chunk counter: chunk g lives in stage buffer g & 1 (two stages) or 0 (one)
  for (;;) {
#pragma unroll
    for (int m = 0; m < MW; ++m)
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        if (ZINIT) {      // lane = pixel, registers 4q..4q+3 = channels 8q + 4h + (0..3) of the 32-block: four 16-byte loads
          typedef __attribute__((ext_vector_type(4))) float f32x4;
          const int gy = cur_y0 + wave * MW + m, gx = cur_x0 + (lane & 31);
          const bool in = gy < H && gx < W;
          const float* zp = a.zinit + (((size_t)cur_n * (a.Cout >> 5) + ((cur_ct * BN + j * 32) >> 5)) * ((size_t)H * W) + (size_t)gy * W + gx) * 32 + 4 * (lane >> 5);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            if (in) z4 = *(const f32x4*)(zp + 8 * q);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][j][4 * q + i] = z4[i];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
        }
      }

    const int next_tile = tile + (int)gridDim.x;
    const bool have_next = next_tile < total_tiles;
    for (int c = 0; c < a.nchunks; ++c, ++g) {
      char* cur = smem + (STAGES == 2 ? (g & 1) : 0) * C::BUF_BYTES;
      const int nxt_halo = (STAGES == 2 ? ((g & 1) ^ 1) : 0) * C::BUF_BYTES;
      const int nxt_slab = nxt_halo + C::HALO_BYTES;
      const bool last = c + 1 == a.nchunks;
      const bool more = !last || have_next;
      const int pc = last ? 0 : c + 1;          // chunk to prefetch (of this tile, or chunk 0 of the next)
      if (last && have_next) {                   // every load of this tile has been issued: switch sources
        int n2, y2, x2, ct2;
        decode(next_tile, n2, y2, x2, ct2);
        setup_sources(n2, y2, x2, ct2);
      }
      // ---- MFMA over 9 taps x KC from the current buffer.  One step = one tap x 16 channels; the
      // fragments of step t+1 are read from LDS (into the other register set) before the MFMAs of
      // step t issue, so the LDS latency hides under the matrix pipe.  The next chunk's global loads
      // (halo items + DMA pieces) are spread over the steps instead of being issued in one burst, and
      // the halo registers are committed to the other buffer just before the last step's MFMAs, so
      // neither the vector-memory issue queue nor the LDS writes idle the pipe.
      const char* halo = cur;
      const char* slab = cur + C::HALO_BYTES;
      constexpr int NSTEPS = 9 * (KC / 16);
#ifndef UNETPP_SPREAD_NW1
#define UNETPP_SPREAD_NW1 3
#endif
      // narrow tiles (NW == 1) finish a chunk in ~3.4k cycles, less than a loaded HBM round trip: their
      // loads go out in the first steps; wide tiles spread them over the whole chunk
      constexpr int SPREAD = (NW == 1) ? UNETPP_SPREAD_NW1 : NSTEPS - 1;
      constexpr int HPS = (ITERS + SPREAD - 1) / SPREAD;          // halo items issued per step
      constexpr int DPS = (DMA_PER_WAVE + SPREAD - 1) / SPREAD;   // DMA pieces issued per step
      if (STAGES == 1) {      // the chunk's requests went out after the previous chunk's MFMAs (or in the prologue)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
      }
      Frag f0, f1;
      load_frags(f0, halo, slab, 0);
#pragma unroll
      for (int st = 0; st < NSTEPS; ++st) {
        Frag& fc = (st & 1) ? f1 : f0;
        Frag& fn = (st & 1) ? f0 : f1;
        if (st + 1 < NSTEPS) load_frags(fn, halo, slab, st + 1);
        if (STAGES == 2 && more) {
#pragma unroll
          for (int k = st * HPS; k < (st + 1) * HPS; ++k)
            if (k < ITERS) halo_dma_one(pc, nxt_halo, k);
#pragma unroll
          for (int k = st * DPS; k < (st + 1) * DPS; ++k)
            if (k < DMA_PER_WAVE) slab_dma_one(pc, nxt_slab, k);
        }
        if (UPF) {
          // chunk c+2's low-res records (landed and published by the end-of-chunk wait + barrier of chunk c+1) ...
          if (st < LSR && c + 2 < a.nchunks && c + 2 >= nch0) ls_dma_one(c + 2, st);
          // ... and chunk c+1's halo image, one item round every UP_EVERY steps, between this chunk's MFMA steps.
          // (Measured: the rounds do not overlap the matrix work of the SIMD's other wave even when the two waves
          // take them at different steps -- all eight waves run in lock-step; see DESIGN.md.)
          constexpr int UP_EVERY = NSTEPS / UPR;
          if (st % UP_EVERY == 0 && st / UP_EVERY < UPR && !last && c + 1 >= nch0) up_item(c + 1, nxt_halo, st / UP_EVERY);
        }
        __builtin_amdgcn_sched_barrier(0);      // keep prefetch + load issue ahead of this step's MFMAs
        run_mfma(fc);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (STAGES == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces have landed
        __syncthreads();
      } else {
        __syncthreads();                                    // every wave has read its last fragments: the stage is free
        if (more) {
#pragma unroll
          for (int k = 0; k < ITERS; ++k) halo_dma_one(pc, 0, k);
#pragma unroll
          for (int k = 0; k < DMA_PER_WAVE; ++k) slab_dma_one(pc, C::HALO_BYTES, k);
        }
      }
    }

    // ---- epilogue straight from registers: scale, bias, ReLU, fp16 (hi/lo) packing, one
    // v_permlane32_swap per word so that every lane owns 8 consecutive channels, 16-byte stores;
    // the fused 2x2 max-pool (unetpp.py:75) is max(row m=0, row m=1) in-lane and one DPP exchange
    // between neighbouring pixel lanes.  No LDS, no barrier: the stores drain under the next tile's MFMAs.
    {
      const int h = lane >> 5;
      const int gx = cur_x0 + (lane & 31);
      const int nbo = a.Cout >> 4;                       // channel blocks of the output tensor
      auto pack_store = [&](const float (&v)[16], half_t* dst, size_t blk_stride, bool ok) { pack_store_octets<P>(v, dst, blk_stride, ok, h); };
#pragma unroll
      for (int j = 0; j < NW; ++j) {
        const int cbase = cur_ct * BN + j * 32;
        float v[MW][16];
        float vmax = 0.f;                                  // largest activation of this lane's part of the tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = cbase + (r & 3) + 8 * (r >> 2) + 4 * h;
          const float2 sb = sb_lds[co];
#pragma unroll
          for (int m = 0; m < MW; ++m) v[m][r] = fmaxf(acc[m][j][r] * sb.x + sb.y, 0.f);
          vmax = (MW == 2) ? fmaxf(fmaxf(vmax, v[0][r]), v[MW - 1][r]) : fmaxf(vmax, v[0][r]);     // v_max3_f32
        }
        // fp16 planes end at 65504: the store below clamps, and says so in the engine's sticky status word.  (A NaN
        // cannot arise here: inputs are sanitised by convert_input, weights are checked when they are loaded.)
        if (!HEAD && __builtin_amdgcn_ballot_w64(vmax > F16_MAX)) {      // rare: report, then clamp to what fp16 can hold
          range_flag(a.status, vmax > F16_MAX, false);
#pragma unroll
          for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int m = 0; m < MW; ++m) v[m][r] = fminf(v[m][r], F16_MAX);
        }
        if (!HEAD) {
#pragma unroll
          for (int m = 0; m < MW; ++m) {
            const int gy = cur_y0 + wave * MW + m;
            const bool ok = gy < H && gx < W;
            const size_t blk = (size_t)H * W * P * 16;
            half_t* dst = a.out + ((size_t)cur_n * nbo + (cbase >> 4)) * blk + ((size_t)gy * W + gx) * P * 16;
            pack_store(v[m], dst, blk, ok);
          }
        } else {
          // logits[c] = b[c] + sum_co x0_4[co] * Wf[c][co] in fp32: 16 channels in this lane, the other 16
          // in lane ^ 32 (one cross-half shuffle); first maximal class wins.
          const size_t hw = (size_t)H * W;
          // class logits of both rows: the 16 head weights a lane needs per class are four float4 in LDS
          // (channels 4h+8q .. +3), read once per class and used for every row of the wave
          float lgm[MW][HEAD_FUSED_MAX_CLASSES];
#pragma unroll
          for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) {
#pragma unroll
            for (int m = 0; m < MW; ++m) lgm[m][c] = -INFINITY;
            if (c < a.head_C) {       // uniform branch
              float wq[16];
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const float4 w4 = *(const float4*)(head_lds + c * 32 + 8 * q + 4 * h);
                wq[4 * q] = w4.x; wq[4 * q + 1] = w4.y; wq[4 * q + 2] = w4.z; wq[4 * q + 3] = w4.w;
              }
              const float bc = head_lds[a.head_C * 32 + c];
#pragma unroll
              for (int m = 0; m < MW; ++m) {
                float part = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) part = fmaf(v[m][r], wq[r], part);
                const float other = __shfl_xor(part, 32, 64);
                lgm[m][c] = bc + (h == 0 ? part + other : other + part);   // same order in both lanes
              }
            }
          }
#pragma unroll
          for (int m = 0; m < MW; ++m) {
            const int gy = cur_y0 + wave * MW + m;
            const bool ok = h == 0 && gy < H && gx < W;
            const size_t pix = (size_t)gy * W + gx;
            float lg[HEAD_FUSED_MAX_CLASSES];
#pragma unroll
            for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c) lg[c] = lgm[m][c];
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
                if (ok && a.probs && c < a.head_C) a.probs[((size_t)cur_n * a.head_C + c) * hw + pix] = pe[c];
              }
              if (a.rule) apply_rule(a.rule, pe[0], pe[1], pe[2], a.t_cable, a.t_tape, a.bg_margin, a.ct_margin, is_cable, is_tape);
            }
            if (ok) {
              if (a.logits) {
#pragma unroll
                for (int c = 0; c < HEAD_FUSED_MAX_CLASSES; ++c)
                  if (c < a.head_C) a.logits[((size_t)cur_n * a.head_C + c) * hw + pix] = lg[c];
              }
              const size_t o = (size_t)cur_n * hw + pix;
              if (a.mask) a.mask[o] = (uint8_t)besti;
              if (a.cable) a.cable[o] = is_cable;
              if (a.tape) a.tape[o] = is_tape;
            }
          }
        }
        if (POOL) {
          static_assert(!POOL || MW == 2, "fused pool needs both rows of a 2x2 window in one wave");
          float pv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float vm = fmaxf(v[0][r], v[MW - 1][r]);
            const float vn = __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, vm), 0xB1, 0xF, 0xF, true));
            pv[r] = fmaxf(vm, vn);       // quad_perm [1,0,3,2]: the horizontally adjacent pixel
          }
          const int Hp = H >> 1, Wp = W >> 1;
          const int py = (cur_y0 >> 1) + wave, px = (cur_x0 >> 1) + ((lane & 31) >> 1);
          const bool ok = ((lane & 1) == 0) && py < Hp && px < Wp;
          const size_t blk = (size_t)Hp * Wp * P * 16;
          half_t* dst = a.pool_out + ((size_t)cur_n * nbo + (cbase >> 4)) * blk + ((size_t)py * Wp + px) * P * 16;
          pack_store(pv, dst, blk, ok);
        }
      }
    }
    if (!have_next) break;
    tile = next_tile;
    decode(tile, cur_n, cur_y0, cur_x0, cur_ct);
  }
}

}  // namespace unetpp
