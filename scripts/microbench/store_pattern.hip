// What the epilogue's store pattern costs.  The conv kernels hold one pixel per lane, so a 16-byte store instruction puts
// lane p's piece at p * 64 + h * 16 (+ 32 for the second plane): 64 lanes touch 32 different 64-byte records, 32 bytes of
// each.  The alternative moves the pieces through LDS first so that lane i writes piece (i & 3) of pixel (i >> 2): one
// contiguous KB per instruction.  Same bytes, same instruction count.
//   per-lane records (as the kernels store today) | coalesced | coalesced after a ds_write_b128 / ds_read_b128 round trip
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sp scripts/microbench/store_pattern.hip && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k_store(char* out, size_t bytes_per_wg, int reps) {
  __shared__ __attribute__((aligned(16))) char stage[8][2560];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = lane & 31, h = lane >> 5;
  char* base = out + (size_t)blockIdx.x * bytes_per_wg;
  const u32x4 v0 = {(unsigned)lane, 1u, 2u, 3u}, v1 = {(unsigned)lane, 5u, 6u, 7u};
  for (int r = 0; r < reps; ++r) {
    // one "row" = 32 pixels x 64-byte records = 2 KB; each wave writes rows wave, wave + 8, ...
    for (size_t row = wave; row * 2048 < bytes_per_wg; row += 8) {
      char* d = base + row * 2048;
      if (MODE == 0) {
        *(u32x4*)(d + p * 64 + h * 16) = v0;
        *(u32x4*)(d + p * 64 + 32 + h * 16) = v1;
      } else if (MODE == 1) {
        *(u32x4*)(d + lane * 16) = v0;
        *(u32x4*)(d + 1024 + lane * 16) = v1;
      } else {
        char* st = stage[wave];
        *(u32x4*)(st + p * 80 + h * 16) = v0;
        *(u32x4*)(st + p * 80 + 32 + h * 16) = v1;
        __builtin_amdgcn_s_waitcnt(0xc07f);                       // lgkmcnt(0)
        const u32x4 a = *(u32x4*)(st + (lane >> 2) * 80 + (lane & 3) * 16);
        const u32x4 b = *(u32x4*)(st + (16 + (lane >> 2)) * 80 + (lane & 3) * 16);
        *(u32x4*)(d + lane * 16) = a;
        *(u32x4*)(d + 1024 + lane * 16) = b;
      }
    }
  }
}

template <int MODE> float run(char* buf, size_t per_wg, int reps) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_store<MODE>, dim3(256), dim3(512), 0, 0, buf, per_wg, 1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_store<MODE>, dim3(256), dim3(512), 0, 0, buf, per_wg, reps);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  for (size_t per_wg : {(size_t)64 << 10, (size_t)1 << 20, (size_t)4 << 20}) {      // 16 MB (stays in L2/MALL), 256 MB, 1 GB in all
    char* buf; (void)hipMalloc((void**)&buf, per_wg * 256);
    const int reps = per_wg >= (1 << 20) ? 4 : 64;
    const double gb = (double)per_wg * 256 * reps / 1e9;
    const float a = run<0>(buf, per_wg, reps), b = run<1>(buf, per_wg, reps), c = run<2>(buf, per_wg, reps);
    printf("%5zu KB per workgroup x %d passes: per-lane records %.3f ms (%.0f GB/s) | coalesced %.3f ms (%.0f GB/s) | through LDS %.3f ms (%.0f GB/s)\n",
           per_wg >> 10, reps, a, gb / a * 1e3, b, gb / b * 1e3, c, gb / c * 1e3);
    (void)hipFree(buf);
  }
  return 0;
}
