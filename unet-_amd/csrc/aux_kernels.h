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
__global__ void weight_scale_kernel(const float* __restrict__ w, int per_co, float* __restrict__ mult_out,
                                    float* __restrict__ scale_out) {
  const int co = blockIdx.x;
  float m = 0.f;
  for (int i = threadIdx.x; i < per_co; i += blockDim.x) m = fmaxf(m, fabsf(w[(size_t)co * per_co + i]));
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

// canonical OIHW fp32 -> [ct][chunk][P][tap][KG][BN][8] fp16 (plane 0 = hi, plane 1 = lo)
__global__ void weight_pack_kernel(const float* __restrict__ w, const float* __restrict__ mult, int Cin, int Cout,
                                   int P, int KC, int BN, int nchunks, half_t* __restrict__ out, long long units) {
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

// ------------------------------------------------------------------------------------------------
// Input conversion -> [N][1][H][W][P][8] fp16 (one channel block of 8, channels 3..7 zero).
//   fmt 0: float32 NCHW RGB in [0,1]                    (model(img_tensor), infer_two_stage_burr.py:292-295)
//   fmt 1: uint8 NHWC BGR: RGB = BGR reversed, /255.0f   (preprocess_image, infer_two_stage_burr.py:122-127)
template <int P>
__global__ void convert_input_kernel(const void* __restrict__ in, int fmt, int N, int H, int W,
                                     half_t* __restrict__ out) {
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
  half8 hi, lo;
#pragma unroll
  for (int e = 0; e < 8; ++e) { hi[e] = (half_t)0.f; lo[e] = (half_t)0.f; }
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    half_t h = (half_t)v[e];
    hi[e] = h;
    lo[e] = (half_t)(v[e] - (float)h);
  }
  half8* o = (half8*)(out + i * P * 8);
  o[0] = hi;
  if (P == 2) o[1] = lo;
}

// ------------------------------------------------------------------------------------------------
// up = nn.Upsample(scale_factor=2, 'bilinear', align_corners=True) (reference unetpp.py:76,112-116).
// src = dst*(in-1)/(out-1) in fp32, i0 = int(src), i1 = i0 + (i0 < in-1), l1 = src - i0, l0 = 1 - l1;
// x is interpolated inside each row first.  The torch.cat([skip, up]) that follows is virtual: the
// decoder conv reads `skip` and this kernel's output as two sources (conv3x3_mfma.h).
template <int P>
__global__ void upsample2x_kernel(const half_t* __restrict__ low, int Cu, int N, int H, int W,
                                  half_t* __restrict__ out) {
  // tensors are channel-blocked: [N][Cu/16][h][w][P][16]; one thread = 8 channels of one output pixel
  const int NBLK = Cu / 16;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * NBLK * H * W * 2;
  if (i >= total) return;
  const int half8i = i & 1;
  size_t p = i >> 1;
  int x = p % W; size_t q = p / W;
  int y = q % H; size_t nb = q / H;          // nb = n * NBLK + channel block
  half_t* dst = out + ((nb * H + y) * W + x) * (size_t)(P * 16) + half8i * 8;
  const int h = H >> 1, w = W >> 1;
  const float sh = h > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f;
  const float sw = w > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
  float fy = sh * (float)y, fx = sw * (float)x;
  int y0 = (int)fy, x0 = (int)fx;
  int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
  float ly1 = fminf(fmaxf(fy - (float)y0, 0.f), 1.f), lx1 = fminf(fmaxf(fx - (float)x0, 0.f), 1.f);
  float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const half_t* b = low + nb * (size_t)h * w * (P * 16) + half8i * 8;
  const half_t* p00 = b + (size_t)(y0 * w + x0) * (P * 16);
  const half_t* p01 = b + (size_t)(y0 * w + x1) * (P * 16);
  const half_t* p10 = b + (size_t)(y1 * w + x0) * (P * 16);
  const half_t* p11 = b + (size_t)(y1 * w + x1) * (P * 16);
  half8 a00 = *(const half8*)p00, a01 = *(const half8*)p01, a10 = *(const half8*)p10, a11 = *(const half8*)p11;
  half8 rh, rl;
  if (P == 2) {
    half8 b00 = *(const half8*)(p00 + 16), b01 = *(const half8*)(p01 + 16);
    half8 b10 = *(const half8*)(p10 + 16), b11 = *(const half8*)(p11 + 16);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v00 = (float)a00[e] + (float)b00[e], v01 = (float)a01[e] + (float)b01[e];
      float v10 = (float)a10[e] + (float)b10[e], v11 = (float)a11[e] + (float)b11[e];
      float v = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
      half_t hi, lo;
      split_f16(v, hi, lo);
      rh[e] = hi; rl[e] = lo;
    }
    *(half8*)dst = rh;
    *(half8*)(dst + 16) = rl;
  } else {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = ly0 * (lx0 * (float)a00[e] + lx1 * (float)a01[e]) + ly1 * (lx0 * (float)a10[e] + lx1 * (float)a11[e]);
      rh[e] = (half_t)v;
    }
    *(half8*)dst = rh;
  }
}

// ------------------------------------------------------------------------------------------------
// self.final = Conv2d(32, C, 1) (unetpp.py:85,119) in fp32, then the frame-loop tail
// softmax -> argmax -> uint8, (pred==1), (pred==2) (infer_two_stage_burr.py:299-304).  softmax is
// monotone, so the class index is taken on the logits; ties resolve to the lowest index (np.argmax).
template <int P>
__global__ void head_argmax_kernel(const half_t* __restrict__ x, const float* __restrict__ w /*[C][32]*/,
                                   const float* __restrict__ b, int C, int N, int H, int W,
                                   float* __restrict__ logits, uint8_t* __restrict__ mask,
                                   uint8_t* __restrict__ cable, uint8_t* __restrict__ tape) {
  __shared__ float ws[HEAD_MAX_CLASSES * 32 + HEAD_MAX_CLASSES];
  for (int i = threadIdx.x; i < C * 32; i += blockDim.x) ws[i] = w[i];
  for (int i = threadIdx.x; i < C; i += blockDim.x) ws[HEAD_MAX_CLASSES * 32 + i] = b[i];
  __syncthreads();
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t hw = (size_t)H * W, total = (size_t)N * hw;
  if (i >= total) return;
  size_t n = i / hw, p = i - n * hw;
  float v[32];
#pragma unroll
  for (int blk = 0; blk < 2; ++blk) {            // x0_4 is channel-blocked: [N][2][H][W][P][16]
    const half8* px = (const half8*)(x + (((n * 2 + blk) * hw + p) * P) * 16);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      half8 hi = px[g];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[blk * 16 + g * 8 + e] = (float)hi[e];
      if (P == 2) {
        half8 lo = px[2 + g];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[blk * 16 + g * 8 + e] += (float)lo[e];
      }
    }
  }
  float best = -INFINITY;
  int besti = 0;
  for (int c = 0; c < C; ++c) {
    float s = ws[HEAD_MAX_CLASSES * 32 + c];
#pragma unroll
    for (int e = 0; e < 32; ++e) s = fmaf(v[e], ws[c * 32 + e], s);
    if (logits) logits[(n * C + c) * hw + p] = s;
    if (s > best) { best = s; besti = c; }
  }
  if (mask) mask[i] = (uint8_t)besti;
  if (cable) cable[i] = besti == 1;
  if (tape) tape[i] = besti == 2;
}

// debug: channel-blocked fp16 -> float32 NCHW
template <int P>
__global__ void unpack_nchw_kernel(const half_t* __restrict__ x, int N, int C, int H, int W, float* __restrict__ out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * C * H * W;
  if (i >= total) return;
  int xx = i % W; size_t t = i / W;
  int y = t % H; t /= H;
  int c = t % C; int n = t / C;
  const int CB = C < 16 ? C : 16;                 // channel-blocked source: [N][C/CB][H][W][P][CB]
  const half_t* p = x + ((((size_t)(n * (C / CB) + c / CB) * H + y) * W + xx) * P) * CB + c % CB;
  float v = (float)p[0];
  if (P == 2) v += (float)p[CB];
  out[i] = v;
}

}  // namespace unetpp
