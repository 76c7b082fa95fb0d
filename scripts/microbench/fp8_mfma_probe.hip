// Probe of the gfx950 pieces a "fp16 main term + fp8 cross terms" mode needs:
//  (1) v_mfma_f32_32x32x16_fp8_fp8: operand layout (same k order as the f16 instruction?) and issue rate vs f16
//  (2) v_cvt_pk_fp8_f32: format (OCP e4m3fn: max 448), rounding, overflow behaviour
//  (3) v_cvt_scalef32_pk_fp8_f32 / _f16: what the scale operand does
// hipcc --offload-arch=gfx950 -O3 -o probe fp8_mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdint>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef short short2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));

__global__ void cvt_probe(const float* in, int n, unsigned* out_pk, unsigned* out_scaled, unsigned* out_f16) {
  int i = threadIdx.x;
  if (i >= n) return;
  float a = in[2 * i], b = in[2 * i + 1];
  out_pk[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
  short2v old = {0, 0};
  short2v r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, 0.25f, false);
  out_scaled[i] = (unsigned)(unsigned short)r[0];
  half2v h = {(_Float16)a, (_Float16)b};
  short2v r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, h, 1.0f, false);
  out_f16[i] = (unsigned)(unsigned short)r2[0];
}

// D = A(32 x 16) * B(16 x 32): operands as the conv kernels hold them: lane l has row/col (l & 31), k = 8 (l >> 5) .. +7
__global__ void mfma_probe(const float* A, const float* B, float* D) {
  const int lane = threadIdx.x;
  long a = 0, b = 0;
  for (int i = 0; i < 8; i += 2) {
    const int k = 8 * (lane >> 5) + i;
    unsigned pa = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(A[(lane & 31) * 16 + k], A[(lane & 31) * 16 + k + 1], 0, false) & 0xffffu;
    unsigned pb = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(B[k * 32 + (lane & 31)], B[(k + 1) * 32 + (lane & 31)], 0, false) & 0xffffu;
    a |= (long)pa << (8 * i);
    b |= (long)pb << (8 * i);
  }
  float16v acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a, b, acc, 0, 0, 0);
  // accumulator layout of the 32x32 f32 result: row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column = lane & 31
  for (int r = 0; r < 16; ++r) D[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
}

typedef int int8v __attribute__((ext_vector_type(8)));
// v_mfma_scale_f32_32x32x64_f8f6f4: FMT 0 = fp8 (e4m3) operands, 2 = fp6, 4 = fp4
template <int FMT>
__global__ __launch_bounds__(256, 1) void rate64(int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  int8v a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38394041 + lane * 3 + i; b[i] = 0x41403938 - lane - i; }
  float16v acc[4] = {};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[m], FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int KIND>
__global__ __launch_bounds__(256, 1) void rate(int iters, unsigned long long* out, float* sink) {
  const int lane = threadIdx.x & 63;
  half8 ah, bh;
  for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(lane * 0.01f + i); bh[i] = (_Float16)(i - lane * 0.02f); }
  long a8 = 0x3839404142434445L + lane, b8 = 0x4544434241403938L - lane;
  float16v acc[4] = {};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      if (KIND == 0) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh, ah, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, ah, acc[m], 0, 0, 0);
      } else if (KIND == 1) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b8, a8, acc[m], 0, 0, 0);
      } else {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, b8, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(b8, a8, acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a8, a8, acc[m], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
  if (s == 12345.f) sink[0] = s;
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

static float e4m3_decode(unsigned b) {
  int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  if (e == 15 && m == 7) v = NAN;
  return s ? -v : v;
}

int main() {
  {
    std::vector<float> in = {1.0f, 0.3f, 448.f, 460.f, 500.f, 1000.f, 1e-3f, 0.0019f, 0.0625f, -2.5f, 17.f, 19.f, 1e9f, INFINITY, 3.14159f, 0.11f};
    float* din; unsigned *d0, *d1, *d2;
    hipMalloc(&din, in.size() * 4); hipMalloc(&d0, 64); hipMalloc(&d1, 64); hipMalloc(&d2, 64);
    hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
    cvt_probe<<<1, 64>>>(din, (int)in.size() / 2, d0, d1, d2);
    unsigned h0[8], h1[8], h2[8];
    hipMemcpy(h0, d0, 32, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, 32, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, 32, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < in.size() / 2; ++i)
      printf("cvt (%g, %g): pk_fp8 -> (%g, %g) [0x%04x]   scalef32(scale .25) -> (%g, %g)   from f16 scale 1 -> (%g, %g)\n", in[2 * i], in[2 * i + 1],
             e4m3_decode(h0[i] & 255), e4m3_decode((h0[i] >> 8) & 255), h0[i] & 0xffff, e4m3_decode(h1[i] & 255), e4m3_decode((h1[i] >> 8) & 255),
             e4m3_decode(h2[i] & 255), e4m3_decode((h2[i] >> 8) & 255));
  }
  {
    std::vector<float> A(32 * 16), B(16 * 32), D(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32; ++i) for (int k = 0; k < 16; ++k) A[i * 16 + k] = (float)((i * 7 + k * 3) % 9 - 4) * 0.25f;     // exactly representable
    for (int k = 0; k < 16; ++k) for (int j = 0; j < 32; ++j) B[k * 32 + j] = (float)((k * 5 + j * 11) % 7 - 3) * 0.5f;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += A[i * 16 + k] * B[k * 32 + j];
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    mfma_probe<<<1, 64>>>(dA, dB, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double err = 0; for (int i = 0; i < 1024; ++i) err = fmax(err, fabs(D[i] - R[i]));
    printf("fp8 mfma 32x32x16 with the f16 kernels' operand and accumulator layout: max |D - ref| = %g (D[0]=%g ref %g, D[37]=%g ref %g)\n", err, D[0], R[0], D[37], R[37]);
  }
  {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 64);
    const int iters = 2000;
    const char* names[3] = {"3 x f16", "f16 + 2 x fp8", "3 x fp8"};
    for (int kind = 0; kind < 3; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) rate<0><<<256, 256>>>(iters, out, sink);
        if (kind == 1) rate<1><<<256, 256>>>(iters, out, sink);
        if (kind == 2) rate<2><<<256, 256>>>(iters, out, sink);
      }
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256);
      hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += v;
      printf("%-14s: %.1f cycles per group of three MFMAs (one wave per SIMD, 4 accumulators)\n", names[kind], s / 256 / (iters * 4.0));
    }
  }
  {
    unsigned long long* out; float* sink;
    hipMalloc(&out, 256 * 8); hipMalloc(&sink, 64);
    const int iters = 2000;
    const char* names[3] = {"fp8", "fp6", "fp4"};
    for (int kind = 0; kind < 3; ++kind) {
      for (int rep = 0; rep < 2; ++rep) {
        if (kind == 0) rate64<0><<<256, 256>>>(iters, out, sink);
        if (kind == 1) rate64<2><<<256, 256>>>(iters, out, sink);
        if (kind == 2) rate64<4><<<256, 256>>>(iters, out, sink);
      }
      hipDeviceSynchronize();
      std::vector<unsigned long long> h(256);
      hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
      double s = 0; for (auto v : h) s += v;
      printf("v_mfma_scale_f32_32x32x64_f8f6f4 %s: %.1f cycles per instruction (K = 64: %.1f per 16 of K)\n", names[kind], s / 256 / (iters * 4.0), s / 256 / (iters * 4.0) / 4);
    }
  }
  return 0;
}
