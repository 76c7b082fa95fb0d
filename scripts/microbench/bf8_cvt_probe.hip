// What the e5m2 conversions used by the EXACT8 planes do on gfx950 (conv3x3_mfma.h split_pack4_x8):
//   v_cvt_scalef32_pk_bf8_f32 d, a, b, scale   (is it a / scale?  rounding?  overflow?)
//   v_cvt_scalef32_f32_bf8    d, bytes, scale  (is it byte * scale?)
// hipcc --offload-arch=gfx950 -O3 -o bf8 bf8_cvt_probe.hip && ./bf8
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef short short2v __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, unsigned* o8, unsigned* om8, float* back) {
  const int i = threadIdx.x;
  if (i >= n) return;
  short2v z = {0, 0};
  short2v r = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(z, in[i], -in[i], 8.0f, false);          // expect e5m2(x / 8)
  o8[i] = (unsigned)(unsigned short)r[0];
  short2v r2 = __builtin_amdgcn_cvt_scalef32_pk_bf8_f32(z, in[i], in[i], 0.00390625f, false);   // expect e5m2(256 x)
  om8[i] = (unsigned)(unsigned short)r2[0];
  back[i] = __builtin_amdgcn_cvt_scalef32_f32_bf8((int)(om8[i] & 255u), 0.00390625f, 0);       // expect ~x again
}
static float e5m2_decode(unsigned b) {
  int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
  float v = e == 0 ? ldexpf(m / 4.f, -14) : (e == 31 ? (m ? NAN : INFINITY) : ldexpf(1.f + m / 4.f, e - 15));
  return s ? -v : v;
}
int main() {
  std::vector<float> in = {0.f, 1.f, 1.125f, 1.375f, 1.625f, 1.875f, 3.3f, 100.f, 65504.f, 8.f * 57344.f, 8.f * 61440.f, 1e-3f, 3e-5f, 6.1e-5f * 8, 1.5e-5f * 8, 0.5e-5f * 8,
                           31.9f, 200.f, 223.9f, 224.1f, 250.f, 1e6f};
  float* din; unsigned *d0, *d1; float* db;
  hipMalloc(&din, 256); hipMalloc(&d0, 256); hipMalloc(&d1, 256); hipMalloc(&db, 256);
  hipMemcpy(din, in.data(), in.size() * 4, hipMemcpyHostToDevice);
  k<<<1, 64>>>(din, (int)in.size(), d0, d1, db);
  std::vector<unsigned> h0(64), h1(64); std::vector<float> hb(64);
  hipMemcpy(h0.data(), d0, 256, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), d1, 256, hipMemcpyDeviceToHost); hipMemcpy(hb.data(), db, 256, hipMemcpyDeviceToHost);
  for (size_t i = 0; i < in.size(); ++i)
    printf("x = %-12g  scale 8: bytes %02x %02x -> %g, %g (x/8 = %g)   scale 2^-8: byte %02x -> %g (256 x = %g)   back through f32_bf8 scale 2^-8: %g\n", in[i],
           h0[i] & 255, (h0[i] >> 8) & 255, e5m2_decode(h0[i] & 255), e5m2_decode((h0[i] >> 8) & 255), in[i] / 8, h1[i] & 255, e5m2_decode(h1[i] & 255), in[i] * 256, hb[i]);
  return 0;
}
