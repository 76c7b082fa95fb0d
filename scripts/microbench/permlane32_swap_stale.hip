// Toolchain finding (hipcc, ROCm 7.2, gfx950): nothing is padded between v_permlane32_swap and a VALU instruction that reads
// its results -- the add below sees the registers as they were BEFORE the swap (sum = 2 * s0 in the low half-wave) unless an
// s_nop 1 sits in between; stores of the swapped words are unaffected.  conv3x3_ws.h's fused head carries that s_nop.
// hipcc --offload-arch=gfx950 -O3 -o swap permlane32_swap_stale.hip && ./swap
#include <hip/hip_runtime.h>
#include <cstdio>
template <bool NOP>
__global__ void k(const float* in, float* out) {
  const int lane = threadIdx.x;
  float p0 = in[lane], p1 = in[64 + lane];
  auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, p0), __builtin_bit_cast(unsigned, p1), false, false);
  unsigned s0 = sw[0], s1 = sw[1];
  if (NOP) asm volatile("s_nop 1" : "+v"(s0), "+v"(s1));
  out[lane] = __builtin_bit_cast(float, s0) + __builtin_bit_cast(float, s1);
}
int main() {
  float h[128], o[64], *di, *dout;
  for (int i = 0; i < 128; ++i) h[i] = (float)(i + 1);
  hipMalloc(&di, 512); hipMalloc(&dout, 256);
  hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
  // after the swap: lanes 0-31 hold (p0[l], p0[l + 32]), lanes 32-63 hold (p1[l - 32], p1[l])
  for (int nop = 0; nop < 2; ++nop) {
    if (nop) k<true><<<1, 64>>>(di, dout); else k<false><<<1, 64>>>(di, dout);
    hipMemcpy(o, dout, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) { const float want = l < 32 ? h[l] + h[l + 32] : h[64 + l - 32] + h[64 + l]; bad += o[l] != want; }
    printf("%s s_nop 1: lane 0 sum = %g (want %g), lane 40 sum = %g (want %g): %d of 64 lanes wrong\n", nop ? "with   " : "without", o[0], h[0] + h[32], o[40],
           h[64 + 8] + h[64 + 40], bad);
  }
  return 0;
}
