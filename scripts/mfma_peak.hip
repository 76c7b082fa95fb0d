// Dev tool: sustained v_mfma_f32_32x32x16_f16 rate of the whole chip for operands held in registers
// (no LDS, no global traffic in the loop), on zero vs random operands: what the power-managed clock gives
// a kernel that does nothing but matrix work.  build: hipcc --offload-arch=gfx950 -O3 -o mfma_peak scripts/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int ORDER>
__global__ __launch_bounds__(512) void mfma_loop(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) % 4096]; b[i] = src[(t * 8 + 4 + i) % 4096]; }
  float16v acc[4] = {};
  for (int it = 0; it < iters; ++it) {
    if (ORDER == 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j) {     // the exact-mode pattern: three MFMAs per accumulator, back to back
          acc[m * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[m + 2], acc[m * 2 + j], 0, 0, 0);
          acc[m * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j + 2], a[m], acc[m * 2 + j], 0, 0, 0);
          acc[m * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[j], a[m], acc[m * 2 + j], 0, 0, 0);
        }
    } else {                              // same work, accumulators interleaved: no two consecutive MFMAs depend
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[m * 2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p == 1 ? b[j + 2] : b[j], p == 0 ? a[m + 2] : a[m], acc[m * 2 + j], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[t] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  half8* src; float* out;
  CK(hipMalloc(&src, 4096 * sizeof(half8))); CK(hipMalloc(&out, 256 * 2 * 512 * sizeof(float)));
  std::vector<_Float16> h(4096 * 8);
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (auto& v : h) v = mode == 0 ? (_Float16)0.f : mode == 1 ? (_Float16)((rand() % 2001 - 1000) / 1000.0f)
                                                    : (rand() % 2 ? (_Float16)((rand() % 1001) / 1000.0f) : (_Float16)0.f);
    CK(hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int order = 0; order < 2; ++order)
    for (int wg = 1; wg <= 2; ++wg) {          // 8 or 16 waves per CU
      auto k = order ? mfma_loop<1> : mfma_loop<0>;
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      hipLaunchKernelGGL(k, dim3(256 * wg), dim3(512), 0, 0, src, out, iters / 10);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(256 * wg), dim3(512), 0, 0, src, out, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flops = 256.0 * wg * 8 * (double)iters * 12 * 32768.0;
      printf("operands %-28s %s %2d waves/CU: %8.3f ms  %7.1f TFLOP/s issued (%.0f %% of 2500)\n",
             mode == 0 ? "zero" : mode == 1 ? "uniform [-1,1]" : "half zero, half [0,1] (ReLU)", order ? "interleaved" : "chained    ", 8 * wg, ms, flops / ms / 1e9,
             flops / ms / 1e9 / 25.0);
    }
  }
  return 0;
}
