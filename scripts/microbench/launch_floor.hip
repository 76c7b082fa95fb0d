// What a dependent launch of the conv kernels' SHAPE costs before it does any work: 256 workgroups x 512 threads, dynamic
// LDS 0 / 64 / 160 KB, (a) empty body, (b) one 8-byte global load per lane -> LDS -> barrier -> one store (the minimal
// "table staged, first barrier passed" prologue), (c) the same after a predecessor that leaves 16 MB of dirty lines.
// Chains of 200 launches on one stream, wall time per launch.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lf scripts/microbench/launch_floor.hip && /tmp/lf
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void k_empty(const float2* in, float* out) {
  extern __shared__ char smem[];
  if (in == nullptr) out[0] = 1.f;
}

__global__ __launch_bounds__(512) void k_touch(const float2* in, float* out) {
  extern __shared__ char smem[];
  float2* t = (float2*)smem;
  t[threadIdx.x] = in[threadIdx.x];
  __syncthreads();
  const float2 v = t[threadIdx.x ^ 1];
  if (v.x == 12345.f) out[blockIdx.x] = v.y;
}

// writes `bytes_per_wg` per workgroup with plain 16-byte stores (left dirty in the XCD L2s at kernel end)
__global__ __launch_bounds__(512) void k_dirty(float4* out, int f4_per_wg) {
  float4* o = out + (size_t)blockIdx.x * f4_per_wg;
  for (int i = threadIdx.x; i < f4_per_wg; i += 512) o[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

template <class F> double chain(hipStream_t s, int n, F f) {
  for (int i = 0; i < 20; ++i) f();
  (void)hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) f();
  (void)hipStreamSynchronize(s);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}

int main() {
  hipStream_t s; (void)hipStreamCreate(&s);
  float2* in; float* out; float4* big;
  (void)hipMalloc((void**)&in, 4096 * 8); (void)hipMemset(in, 0, 4096 * 8);
  (void)hipMalloc((void**)&out, 4096 * 4);
  (void)hipMalloc((void**)&big, 64u << 20);
  for (auto k : {k_empty, k_touch}) (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int wgs : {256, 1024}) for (int lds : {0, 65536, 163840}) {
    const double e = chain(s, 200, [&] { hipLaunchKernelGGL(k_empty, dim3(wgs), dim3(512), lds, s, in, out); });
    const double t = chain(s, 200, [&] { hipLaunchKernelGGL(k_touch, dim3(wgs), dim3(512), lds, s, in, out); });
    printf("%4d workgroups x 512 threads, %3d KB LDS: empty %.2f us per launch, staged prologue %.2f us\n", wgs, lds >> 10, e, t);
  }
  for (int mb : {0, 2, 16, 32}) {
    const int f4 = mb ? (mb << 20) / 16 / 256 : 0;
    const double d = chain(s, 200, [&] {
      if (mb) hipLaunchKernelGGL(k_dirty, dim3(256), dim3(512), 0, s, big, f4);
      hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 163840, s, in, out);
    });
    const double a = mb ? chain(s, 200, [&] { hipLaunchKernelGGL(k_dirty, dim3(256), dim3(512), 0, s, big, f4); }) : 0.0;
    printf("predecessor leaves %2d MB dirty: pair %.2f us, writer alone %.2f us\n", mb, d, a);
  }
  return 0;
}
