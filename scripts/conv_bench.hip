// Dev tool (not part of the product): times one conv3x3 configuration on random data.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o build/conv_bench scripts/conv_bench.hip
// usage: [CB_DATA=0|1|2] conv_bench P KC NW N H W C0 C1 Cout [reps [MW [WAVES]]]
//   CB_DATA: operand values 0 = zeros, 1 = uniform [-1,1] (default), 2 = ReLU-like (half zeros, half [0,1]) - the rate the
//   chip sustains depends on them
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "../unet-_amd/csrc/conv3x3_mfma.h"
using namespace unetpp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int P, int KC, int NW, int MW, int WV>
float run(ConvArgs a, int reps) {
  using C = ConvCfg<P, KC, NW, MW, WV>;
  auto k = conv3x3_bias_relu_kernel<P, KC, NW, MW, WV, false>;
  CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES + 8192));
  a.tiles_x = (a.W + 31) / 32; a.tiles_y = (a.H + C::TH - 1) / C::TH; a.nct = a.Cout / C::BN;
  a.nchunks = (a.C0 + KC - 1) / KC + a.C1 / KC;
  int total = a.N * a.tiles_x * a.tiles_y * a.nct;
  int per_cu = std::max(1, std::min(2, (160 * 1024) / C::LDS_BYTES));
  dim3 grid(std::min(total, 256 * per_cu));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, grid, dim3(C::NT), C::LDS_BYTES + a.Cout * 8, 0, a);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, grid, dim3(C::NT), C::LDS_BYTES + a.Cout * 8, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("grid %u LDS %d\n", grid.x, C::LDS_BYTES);
  return ms / reps;
}

int main(int argc, char** argv) {
  if (argc < 10) { printf("usage\n"); return 1; }
  int P = atoi(argv[1]), KC = atoi(argv[2]), NW = atoi(argv[3]);
  ConvArgs a{};
  a.N = atoi(argv[4]); a.H = atoi(argv[5]); a.W = atoi(argv[6]); a.C0 = atoi(argv[7]); a.C1 = atoi(argv[8]); a.Cout = atoi(argv[9]);
  int reps = argc > 10 ? atoi(argv[10]) : 20;
  int MW = argc > 11 ? atoi(argv[11]) : 2, WV = argc > 12 ? atoi(argv[12]) : 8;
  size_t px = (size_t)a.N * a.H * a.W;
  size_t n0 = px * P * a.C0, n1 = px * P * (a.C1 ? a.C1 : 8), no = px * P * a.Cout;
  size_t nw = (size_t)(a.C0 + a.C1 + 32) * 9 * a.Cout * P + 65536;
  std::vector<half_t> h(std::max(std::max(n0, n1), nw));
  srand(1);
  { const char* z = getenv("CB_DATA"); const int mode = z ? atoi(z) : 1;
    for (auto& v : h) v = mode == 0 ? (half_t)0.f : mode == 1 ? (half_t)((rand() % 2001 - 1000) / 1000.0f)
                                   : (rand() % 2 ? (half_t)((rand() % 1001) / 1000.0f) : (half_t)0.f); }
  half_t *d0, *d1, *dw, *dout; float *sc, *bi;
  CK(hipMalloc(&d0, n0 * 2)); CK(hipMalloc(&d1, n1 * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dout, no * 2));
  CK(hipMalloc(&sc, a.Cout * 4)); CK(hipMalloc(&bi, a.Cout * 4));
  CK(hipMemcpy(d0, h.data(), n0 * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(d1, h.data(), n1 * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(dw, h.data(), nw * 2, hipMemcpyHostToDevice));
  std::vector<float> ones(a.Cout, 1e-3f); CK(hipMemcpy(sc, ones.data(), a.Cout * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bi, ones.data(), a.Cout * 4, hipMemcpyHostToDevice));
  a.in0 = d0; a.in1 = a.C1 ? d1 : nullptr; a.wpk = dw; a.scale = sc; a.bias = bi; a.out = dout; a.pool_out = nullptr;
  float ms = -1;
#define CASE(p, kc, nw_, mw, wv) if (P == p && KC == kc && NW == nw_ && MW == mw && WV == wv) ms = run<p, kc, nw_, mw, wv>(a, reps);
  CASE(2, 16, 2, 2, 8) CASE(2, 16, 1, 2, 8) CASE(1, 16, 4, 2, 8) CASE(1, 32, 2, 2, 8) CASE(1, 32, 1, 2, 8)
  CASE(2, 16, 1, 2, 4) CASE(2, 16, 2, 2, 4) CASE(1, 32, 1, 2, 4) CASE(1, 32, 2, 2, 4) CASE(2, 16, 1, 4, 4) CASE(1, 16, 1, 2, 4) CASE(1, 16, 2, 2, 4) CASE(1, 32, 1, 4, 4) CASE(1, 16, 1, 4, 8) CASE(1, 16, 1, 4, 4) CASE(1, 16, 1, 2, 8)
  double fl = 2.0 * px * a.Cout * (a.C0 + a.C1) * 9;
  printf("MW%d WV%d P%d KC%d NW%d N%d %dx%d C0=%d C1=%d Cout=%d : %.1f us  %.1f TF/s alg (%.1f TF/s MFMA)\n", MW, WV, P, KC, NW, a.N, a.H, a.W, a.C0, a.C1,
         a.Cout, ms * 1e3, fl / ms / 1e9, fl * (P == 2 ? 3 : 1) / ms / 1e9);
  return 0;
}
