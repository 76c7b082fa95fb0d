// What the EXACT8 product  x*w = xh*wh [fp16 MFMA] + (xl8*wh8 + x8*wl8) [ONE v_mfma_scale_f32_32x32x64_f8f6f4, the two
// cross terms concatenated along K]  needs to know about gfx950, measured:
//  (1) operand layout of v_mfma_scale_f32_32x32x64_f8f6f4 (found with scale_mfma_layout.hip): byte b of lane l of one operand
//      meets byte b of the lane with the same l >> 5 of the other; the E8M0 scale of lane r < 32 covers bytes 0..15 of lanes r
//      and r + 32, that of lane r + 32 their bytes 16..31; mixed formats (first operand e4m3, second e5m2); accumulator layout
//  (2) sustained whole-chip rate (wall time, random operands, one MFMA wave per SIMD as in conv3x3_ws_kernel) of the
//      instruction mix per (tap, 32 input channels, 32x32 output block):
//         exact:   6 x v_mfma_f32_32x32x16_f16                       (192 pipe cycles)
//         exact8:  2 x v_mfma_f32_32x32x16_f16 + 1 x scale-f8 K=64   (128 pipe cycles)
//         exact6:  2 x f16 + 1 x scale-f6 K=64                       ( 96 pipe cycles)
// hipcc --offload-arch=gfx950 -O3 -o cross8 cross8_mfma.hip && ./cross8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstdint>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef int int8v __attribute__((ext_vector_type(8)));

static float e4m3_decode(unsigned b) {
  int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
  return s ? -v : v;
}
static float e5m2_decode(unsigned b) {
  int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
  float v = e == 0 ? ldexpf(m / 4.f, -14) : ldexpf(1.f + m / 4.f, e - 15);
  return s ? -v : v;
}

// lane l passes its 32 bytes of each operand as they lie in memory at [l][0..31], scale bytes from sa[l], sb[l]
__global__ void layout_probe(const unsigned char* A, const unsigned char* B, const unsigned char* sa, const unsigned char* sb, float* D) {
  const int lane = threadIdx.x;
  int8v a, b;
  for (int i = 0; i < 8; ++i) { a[i] = ((const int*)(A + lane * 32))[i]; b[i] = ((const int*)(B + lane * 32))[i]; }
  float16v acc = {};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0 /* first: e4m3 */, 1 /* second: e5m2 */, 0, (int)sa[lane], 0, (int)sb[lane]);
  for (int r = 0; r < 16; ++r) D[lane * 16 + r] = acc[r];
}

template <int KIND>
__global__ __launch_bounds__(256, 1) void mix_loop(const int* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  half8 ah[2], bh[2], al[2], bl[2];
  int8v a8[2], b8[2];
  const int* p = src + (size_t)(t % 2048) * 64;
  for (int i = 0; i < 2; ++i) {
    ah[i] = *(const half8*)(p + 4 * i); bh[i] = *(const half8*)(p + 8 + 4 * i);
    al[i] = *(const half8*)(p + 16 + 4 * i); bl[i] = *(const half8*)(p + 24 + 4 * i);
    for (int k = 0; k < 8; ++k) { a8[i][k] = p[32 + 8 * i + k] & 0x77777777; b8[i][k] = p[48 + 8 * i + k] & 0x73737373; }   // no NaN / Inf encodings
  }
  float16v acc[4] = {};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      // main term of 32 channels: two K = 16 steps
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m & 1], bh[m >> 1], acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(m & 1) ^ 1], bh[(m >> 1) ^ 1], acc[m], 0, 0, 0);
      if (KIND == 0) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m & 1], bl[m >> 1], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m & 1], bh[m >> 1], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(m & 1) ^ 1], bl[(m >> 1) ^ 1], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[(m & 1) ^ 1], bh[(m >> 1) ^ 1], acc[m], 0, 0, 0);
      } else if (KIND == 1) {
        acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[m & 1], b8[m >> 1], acc[m], 0, 1, 0, 0x7a7b7c7d, 0, 0x7f7f7f7f);
      } else if (KIND == 2) {
        acc[m] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[m & 1], b8[m >> 1], acc[m], 2, 2, 0, 0x7a7b7c7d, 0, 0x7f7f7f7f);
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

int main(int argc, char** argv) {
  {
    // ---- (1) layout
    std::vector<unsigned char> A(64 * 32), B(64 * 32), sa(64), sb(64);
    srand(7);
    for (auto& v : A) { v = rand() & 0xff; if ((v & 0x7f) == 0x7f) v ^= 1; }               // e4m3: 0x7f / 0xff are NaN
    for (auto& v : B) { v = rand() & 0xff; if (((v >> 2) & 31) == 31) v &= 0xbf; }          // e5m2: exponent 31 is Inf / NaN
    for (int l = 0; l < 64; ++l) { sa[l] = 124 + rand() % 7; sb[l] = 125 + rand() % 5; }
    unsigned char *dA, *dB, *dsa, *dsb; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 64 * 16 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
    layout_probe<<<1, 64>>>(dA, dB, dsa, dsb, dD);
    std::vector<float> D(64 * 16);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    double err = 0, mag = 0;
    for (int lane = 0; lane < 64; ++lane)
      for (int r = 0; r < 16; ++r) {
        const int i = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), j = lane & 31;       // accumulator layout of the f16 instruction
        double ref = 0;
        // measured (scale_mfma_layout.hip): byte b of lane (row | 32 hh) of one operand meets byte b of lane (col | 32 hh) of
        // the other; the scale byte of lane `row` covers bytes 0..15 of lanes row and row + 32, that of lane row + 32 bytes 16..31
        for (int hh = 0; hh < 2; ++hh)
          for (int k = 0; k < 32; ++k)
            ref += (double)e4m3_decode(A[(i + 32 * hh) * 32 + k]) * (double)e5m2_decode(B[(j + 32 * hh) * 32 + k]) *
                   ldexp(1.0, (int)sa[i + 32 * (k >> 4)] - 127) * ldexp(1.0, (int)sb[j + 32 * (k >> 4)] - 127);
        err = fmax(err, fabs(D[lane * 16 + r] - ref)); mag = fmax(mag, fabs(ref));
      }
    printf("layout probe (first operand e4m3 + per-lane scale, second e5m2 + per-lane scale): max |D - ref| = %g of max |ref| = %g\n", err, mag);
  }
  {
    const int iters = argc > 1 ? atoi(argv[1]) : 60000;
    int* src; float* out;
    hipMalloc(&src, 2048 * 64 * 4); hipMalloc(&out, 256 * 256 * 4);
    std::vector<int> h(2048 * 64);
    srand(3);
    for (size_t i = 0; i < h.size(); ++i) {
      if ((i % 64) < 32) {                        // fp16 pairs in (-1, 1), a third of them zero (ReLU-like)
        _Float16 v[2];
        for (int k = 0; k < 2; ++k) v[k] = rand() % 3 == 0 ? (_Float16)0.f : (_Float16)((rand() % 2001 - 1000) / 1000.0f);
        h[i] = *(int*)v;
      } else h[i] = rand() ^ (rand() << 16);
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[3] = {"exact  (6 x f16)", "exact8 (2 x f16 + f8 K=64)", "exact6 (2 x f16 + f6 K=64)"};
    const double cyc[3] = {192, 128, 96};
    for (int round = 0; round < 3; ++round)
      for (int kind = 0; kind < 3; ++kind) {
        hipEventRecord(e0);
        if (kind == 0) mix_loop<0><<<256, 256>>>(src, out, iters);
        if (kind == 1) mix_loop<1><<<256, 256>>>(src, out, iters);
        if (kind == 2) mix_loop<2><<<256, 256>>>(src, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double units = 256.0 * 4 * 4 * iters;           // (tap, 32 channels, 32x32 block) products, whole chip
        printf("round %d  %-28s %8.2f ms  %.2f ns per unit per SIMD  -> pipe clock %.2f GHz at %3.0f cycles/unit;  useful rate %.0f TFLOP/s\n", round, names[kind], ms,
               ms * 1e6 / (4.0 * iters), cyc[kind] / (ms * 1e6 / (4.0 * iters)), cyc[kind], units * 2.0 * 32 * 32 * 32 / (ms * 1e-3) / 1e12);
      }
  }
  return 0;
}
