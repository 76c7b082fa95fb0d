// What the e4m3 conversions a more accurate exact8 would use do on gfx950:
//   v_cvt_scalef32_pk_fp8_f32 d, a, b, scale   (a / scale?  rounding?  what happens beyond 448?)
//   v_cvt_scalef32_f32_fp8    d, byte, scale
//   v_cvt_scalef32_pk_f16_fp8 d, bytes, scale
// hipcc --offload-arch=gfx950 -O3 -o fp8 fp8_cvt_probe.hip && ./fp8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef short short2v __attribute__((ext_vector_type(2)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, unsigned* o1, unsigned* o13, float* back, float* backh) {
  const int i = threadIdx.x;
  if (i >= n) return;
  short2v z = {0, 0};
  short2v r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, in[i], -in[i], 1.0f, false);            // expect e4m3(x)
  o1[i] = (unsigned)(unsigned short)r[0];
  short2v r2 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(z, in[i], in[i], 0.0001220703125f, false);  // expect e4m3(2^13 x)
  o13[i] = (unsigned)(unsigned short)r2[0];
  back[i] = __builtin_amdgcn_cvt_scalef32_f32_fp8((int)(o13[i] & 255u), 0.0001220703125f, 0);      // expect ~x again
  half2v hh = __builtin_amdgcn_cvt_scalef32_pk_f16_fp8((int)(o1[i] & 0xffffu), 1.0f, false);
  backh[i] = (float)hh[0];
}
static float e4m3_decode(unsigned b) {
  int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
  float v = e == 0 ? ldexpf(m / 8.f, -6) : ((e == 15 && m == 7) ? NAN : ldexpf(1.f + m / 8.f, e - 7));
  return s ? -v : v;
}
int main() {
  std::vector<float> in = {0.f, 1.f, 1.0625f, 1.1875f, 3.3f, 100.f, 440.f, 448.f, 463.9f, 464.1f, 480.f, 1000.f, 65504.f, INFINITY, NAN, 0.0156f, 0.004f, 0.00098f, 0.0005f,
                           0.03f, 0.054f, 0.055f, 0.06f, 1e-6f, 2.4e-7f, 1e-7f};
  float* din; unsigned *d0, *d1; float *db, *dh;
  hipMalloc(&din, 256); hipMalloc(&d0, 256); hipMalloc(&d1, 256); hipMalloc(&db, 256); hipMalloc(&dh, 256);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, (int)in.size(), d0, d1, db, dh);
  std::vector<unsigned> h0(64), h1(64); std::vector<float> hb(64), hh(64);
  hipMemcpy(h0.data(), d0, 256, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), d1, 256, hipMemcpyDeviceToHost);
  hipMemcpy(hb.data(), db, 256, hipMemcpyDeviceToHost); hipMemcpy(hh.data(), dh, 256, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < in.size(); ++i)
    printf("x = %-12g scale 1: bytes %02x %02x -> %g, %g; through pk_f16_fp8: %g   |   scale 2^-13: byte %02x -> %g (2^13 x = %g), back through f32_fp8: %g\n", in[i],
           h0[i] & 255, (h0[i] >> 8) & 255, e4m3_decode(h0[i] & 255), e4m3_decode((h0[i] >> 8) & 255), hh[i], h1[i] & 255, e4m3_decode(h1[i] & 255), in[i] * 8192, hb[i]);
  return 0;
}
