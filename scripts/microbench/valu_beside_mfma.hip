// How fast does a second wave's VALU stream issue on a SIMD whose first wave issues v_mfma_f32_32x32x16_f16 back to back?
// One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run MFMAs (mode bit 0 on), waves 4-7 run a straight-line
// VALU block of one instruction kind and report cycles per instruction.  hipcc --offload-arch=gfx950 -O3 -o vbm valu_beside_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));

template <int KIND>
__device__ __forceinline__ void valu_block(float (&x)[16], float w0, float w1) {
#pragma unroll
  for (int rep = 0; rep < 4; ++rep)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(w0), "v"(w1));
      if (KIND == 1) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(*(float2v*)&x[i & ~1]) : "v"(*(float2v*)&x[(i & ~1) ^ 2]));
      if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
      if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
      if (KIND == 4) asm volatile("v_add_u32 %0, %1, %0" : "+v"(x[i]) : "v"(w0));
      if (KIND == 5) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(x[i]), "+v"(x[(i + 1) & 15]));
      if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(x[i]) : "v"(x[(i + 1) & 15]));
      if (KIND == 7) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
      if (KIND == 8) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(w1), "v"(w0));
      if (KIND == 9) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]), "v"(w0));
      if (KIND == 10) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
      if (KIND == 11) asm volatile("v_exp_f32 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 15]));
      if (KIND == 12) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]) : "vcc");
      if (KIND == 13) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(x[i]) : "v"(x[(i + 1) & 15]));
      if (KIND == 14) asm volatile("v_pk_mul_f16 %0, %1, %2" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
      if (KIND == 15) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
      if (KIND == 16) asm volatile("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(x[i]) : "v"(x[(i + 1) & 15]), "v"(x[(i + 2) & 15]));
      if (KIND == 17) asm volatile("v_readlane_b32 s20, %1, 3\n\tv_mov_b32 %0, s20" : "=v"(x[i]) : "v"(x[(i + 1) & 15]) : "s20");
    }
}

template <int KIND>
__global__ __launch_bounds__(512, 1) void k(int mode, int iters, unsigned long long* out, float* sink) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < 4) {
    if (!(mode & 1)) return;
    if (mode & 4) __builtin_amdgcn_s_setprio(0);
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane * 0.01f + i); b[i] = (_Float16)(i - lane * 0.02f); }
    float16v acc[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
        if (mode & 2) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[m], 0, 0, 0);
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, a, acc[m], 0, 0, 0);
        }
      }
    }
    float s = 0;
    for (int m = 0; m < 4; ++m) for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 12345.f) sink[0] = s;
    return;
  }
  if (mode & 4) __builtin_amdgcn_s_setprio(3);
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = lane + i;
  const float w0 = 1.0001f, w1 = 0.5f;
  const int viters = iters / 8;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < viters; ++it) valu_block<KIND>(x, w0, w1);
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  if (s == 12345.f) sink[1] = s;
  if (lane == 0) out[blockIdx.x * 4 + (wave - 4)] = t1 - t0;
}

template <int KIND>
void run(const char* name, int mode) {
  unsigned long long* out; float* sink;
  hipMalloc(&out, 256 * 4 * 8); hipMalloc(&sink, 64);
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) k<KIND><<<256, 512>>>(mode, iters, out, sink);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(1024);
  hipMemcpy(h.data(), out, 1024 * 8, hipMemcpyDeviceToHost);
  double s = 0; for (auto v : h) s += v;
  const double per = s / 1024 / ((iters / 8) * 64.0);
  printf("%-18s mfma=%d chain3=%d prio=%d : %.2f cycles per VALU instruction\n", name, mode & 1, (mode >> 1) & 1, (mode >> 2) & 1, per);
  hipFree(out); hipFree(sink);
}

int main() {
  for (int mode : {0, 1, 3, 7}) {
    run<0>("v_fma_f32", mode);
    run<1>("v_pk_fma_f32", mode);
    run<2>("v_fma_mix_f32", mode);
    run<3>("v_cvt_pk_f16_f32", mode);
    run<4>("v_add_u32", mode);
    if (mode == 0 || mode == 3) {
      run<5>("v_permlane32_swap", mode); run<6>("v_mov_b32 dpp", mode); run<7>("v_max3_f32", mode); run<8>("v_med3_f32", mode);
      run<9>("v_perm_b32", mode); run<10>("v_fma_mixlo_f16", mode); run<11>("v_exp_f32", mode); run<12>("v_cndmask_b32", mode);
      run<13>("v_cvt_f32_f16", mode); run<14>("v_pk_mul_f16", mode); run<15>("v_mul_lo_u32", mode); run<16>("v_lshl_add_u32", mode);
      run<17>("v_readlane+v_mov", mode);
    }
  }
  return 0;
}
