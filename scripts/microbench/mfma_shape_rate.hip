// Sustained whole-chip rate of the two fp16 MFMA shapes on the same operand data (registers only, 8 waves per CU, three
// MFMAs chained per accumulator as in the exact-mode convolution): does the power-managed clock favour one shape?
// hipcc --offload-arch=gfx950 -O3 -o shape mfma_shape_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

// the same comparison without dependent chains: every accumulator is touched once per pass over all of them
template <int SHAPE>
__global__ __launch_bounds__(512) void loop_il(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) % 4096]; b[i] = src[(t * 8 + 4 + i) % 4096]; }
  float s = 0.f;
  if (SHAPE == 0) {
    float16v acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < 4; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(p == 1 ? b[(m & 1) + 2] : b[m & 1], p == 0 ? a[(m >> 1) + 2] : a[m >> 1], acc[m], 0, 0, 0);
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    float4v acc[8] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int m = 0; m < 8; ++m)
          acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(p == 1 ? b[(m & 1) + 2] : b[m & 1], p == 0 ? a[((m >> 1) & 1) + 2] : a[(m >> 1) & 1], acc[m], 0, 0, 0);
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[t] = s;
}

template <int SHAPE>
__global__ __launch_bounds__(512) void loop(const half8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  half8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 8 + i) % 4096]; b[i] = src[(t * 8 + 4 + i) % 4096]; }
  float s = 0.f;
  if (SHAPE == 0) {
    float16v acc[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[m & 1], a[(m >> 1) + 2], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[(m & 1) + 2], a[m >> 1], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[m & 1], a[m >> 1], acc[m], 0, 0, 0);
      }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    float4v acc[8] = {};                       // the same flops per trip: 16x16x32 is half a 32x32x16
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[m & 1], a[((m >> 1) & 1) + 2], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[(m & 1) + 2], a[(m >> 1) & 1], acc[m], 0, 0, 0);
        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[m & 1], a[(m >> 1) & 1], acc[m], 0, 0, 0);
      }
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  out[t] = s;
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  half8* src; float* out;
  hipMalloc(&src, 4096 * sizeof(half8)); hipMalloc(&out, 256 * 512 * sizeof(float));
  std::vector<_Float16> h(4096 * 8);
  for (int mode = 0; mode < 3; ++mode) {
    srand(1);
    for (auto& v : h) v = mode == 0 ? (_Float16)0.f : mode == 1 ? (_Float16)((rand() % 2001 - 1000) / 1000.0f)
                                                    : (rand() % 2 ? (_Float16)((rand() % 1001) / 1000.0f) : (_Float16)0.f);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    for (int variant = 0; variant < 3; ++variant)      // 0: chained, 8 waves/CU; 1: interleaved, 8 waves/CU; 2: chained, 4 waves/CU (one per SIMD)
    for (int shape = 0; shape < 2; ++shape) {
      auto k = variant == 1 ? (shape ? loop_il<1> : loop_il<0>) : (shape ? loop<1> : loop<0>);
      const int threads = variant == 2 ? 256 : 512;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, src, out, iters / 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, src, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = 256.0 * (threads / 64) * (double)iters * 12 * 32768.0;
      printf("operands %-28s %-10s %-26s: %8.3f ms  %7.1f TFLOP/s issued (%.0f %% of 2500)\n",
             mode == 0 ? "zero" : mode == 1 ? "uniform [-1,1]" : "half zero, half [0,1] (ReLU)", shape ? "16x16x32" : "32x32x16",
             variant == 0 ? "chained x3, 2 waves/SIMD" : variant == 1 ? "interleaved, 2 waves/SIMD" : "chained x3, 1 wave/SIMD", ms, flops / ms / 1e9, flops / ms / 1e9 / 25.0);
    }
  }
  return 0;
}
