// What does one LDS-DMA instruction (buffer_load_dwordx4 ... lds, 1 KiB per wave) cost a loader wave, depending on where
// the bytes come from and how the 64 lanes address them?  256 workgroups x 256 threads (4 loader waves per CU, as in the
// conv kernel), each wave issues BATCH pieces, waits, repeats.  hipcc --offload-arch=gfx950 -O3 -o dma lds_dma_issue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void blds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, int soff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
// mode 0: every workgroup re-reads its own 64 KiB window (L2 resident after the first pass)
// mode 1: streaming: every piece a new KiB of a 1 GiB buffer, pieces of one wave contiguous
// mode 2: halo-like: piece p = 16 pixels x 64 B of image row (p), rows 32 KiB apart, a new tile every batch
template <int BATCH>
__global__ __launch_bounds__(256, 1) void k(const char* src, size_t bytes, int mode, int iters, unsigned long long* out, int soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
  unsigned long long issue = 0, wait = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll
    for (int p = 0; p < BATCH; ++p) {
      size_t off;
      if (mode == 0) off = ((size_t)blockIdx.x * 65536 + (size_t)((wave * BATCH + p) % 64) * 1024) % (bytes - 65536);
      else if (mode == 1) off = (((size_t)it * gridDim.x + blockIdx.x) * 4 * BATCH + wave * BATCH + p) * 1024 % (bytes - 65536);
      else off = ((((size_t)it * gridDim.x + blockIdx.x) * 2048) % 32768 + (size_t)(wave * BATCH + p) * 32768 + ((size_t)((it * gridDim.x + blockIdx.x) / 16) % 400) * 2097152) % (bytes - 65536);
      const unsigned base = (unsigned)(off & 0x7fffffffu);      // (1 GiB buffer: offsets fit)
      blds16(rs, base + lane * 16, soff, lds_base + (wave * BATCH + p) * 1024);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_readcyclecounter();
    issue += t1 - t0; wait += t2 - t1;
    __syncthreads();
  }
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = issue; out[(blockIdx.x * 4 + wave) * 2 + 1] = wait; }
}
// the same three sources through registers: buffer_load_dwordx4 into VGPRs, then ds_write_b128
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int BATCH>
__global__ __launch_bounds__(256, 1) void kreg(const char* src, size_t bytes, int mode, int iters, unsigned long long* out, int soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
  unsigned long long issue = 0, wait = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned long long t0 = __builtin_readcyclecounter();
    u32x4 v[BATCH];
#pragma unroll
    for (int p = 0; p < BATCH; ++p) {
      size_t off;
      if (mode == 0) off = ((size_t)blockIdx.x * 65536 + (size_t)((wave * BATCH + p) % 64) * 1024) % (bytes - 65536);
      else if (mode == 1) off = (((size_t)it * gridDim.x + blockIdx.x) * 4 * BATCH + wave * BATCH + p) * 1024 % (bytes - 65536);
      else off = ((((size_t)it * gridDim.x + blockIdx.x) * 2048) % 32768 + (size_t)(wave * BATCH + p) * 32768 + ((size_t)((it * gridDim.x + blockIdx.x) / 16) % 400) * 2097152) % (bytes - 65536);
      const unsigned base = (unsigned)(off & 0x7fffffffu);
      v[p] = __builtin_amdgcn_raw_buffer_load_b128(rs, base + lane * 16, soff, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
    for (int p = 0; p < BATCH; ++p) *(u32x4*)(smem + (wave * BATCH + p) * 1024 + lane * 16) = v[p];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t2 = __builtin_readcyclecounter();
    issue += t1 - t0; wait += t2 - t1;
    __syncthreads();
  }
  if (lane == 0) { out[(blockIdx.x * 4 + wave) * 2] = issue; out[(blockIdx.x * 4 + wave) * 2 + 1] = wait; }
}
// how the L2-resident rate scales with the number of loader waves per CU (1 .. 8 of a 512-thread workgroup)
__global__ __launch_bounds__(512, 1) void kwaves(const char* src, size_t bytes, int nwaves, int iters, unsigned long long* out, int soff) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0x7fffffff, 0x00020000);
  unsigned long long total = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned long long t0 = __builtin_readcyclecounter();
    if (wave < nwaves) {
#pragma unroll
      for (int p = 0; p < 10; ++p) {
        const size_t off = ((size_t)blockIdx.x * 131072 + (size_t)((wave * 10 + p) % 96) * 1024) % (bytes - 131072);
        blds16(rs, (unsigned)off + lane * 16, soff, lds_base + (wave * 10 + p) * 1024);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    total += __builtin_readcyclecounter() - t0;
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = total;
}
int main() {
  const size_t bytes = 1ull << 30;
  char* src; unsigned long long* out;
  hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&out, 256 * 4 * 2 * 8);
  const int iters = 200;
  const char* names[3] = {"L2-resident window", "streaming from HBM", "halo-like rows 32 KiB apart"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) k<10><<<256, 256, 160 * 1024 - 1024>>>(src, bytes, mode, iters, out, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2048);
    hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost);
    double is = 0, wt = 0; for (int i = 0; i < 1024; ++i) { is += h[2 * i]; wt += h[2 * i + 1]; }
    is /= 1024.0 * iters; wt /= 1024.0 * iters;
    printf("%-28s: 10 pieces per wave and batch: issue %.0f cycles (%.0f per piece), then wait %.0f; %.1f KB per us and CU at 2.0 GHz\n", names[mode], is, is / 10, wt,
           40.0 / ((is + wt) / 2000.0));
  }
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) kreg<10><<<256, 256, 160 * 1024 - 1024>>>(src, bytes, mode, iters, out, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(2048);
    hipMemcpy(h.data(), out, 2048 * 8, hipMemcpyDeviceToHost);
    double is = 0, wt = 0; for (int i = 0; i < 1024; ++i) { is += h[2 * i]; wt += h[2 * i + 1]; }
    is /= 1024.0 * iters; wt /= 1024.0 * iters;
    printf("via registers, %-28s: issue %.0f cycles (%.0f per piece), then ds_write + wait %.0f; %.1f KB per us and CU at 2.0 GHz\n", names[mode], is, is / 10, wt,
           40.0 / ((is + wt) / 2000.0));
  }
  for (int nw = 1; nw <= 8; nw *= 2) {
    for (int rep = 0; rep < 2; ++rep) kwaves<<<256, 512, 160 * 1024 - 1024>>>(src, bytes, nw, iters, out, 0);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(256);
    hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost);
    double t = 0; for (auto v : h) t += v;
    t /= 256.0 * iters;
    printf("L2-resident LDS-DMA, %d loader waves per CU: %.0f cycles per batch of %d KB -> %.1f KB per us and CU at 2.0 GHz\n", nw, t, nw * 10, nw * 10 / (t / 2000.0));
  }
  return 0;
}
