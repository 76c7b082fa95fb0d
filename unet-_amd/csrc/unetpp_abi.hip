// unetpp_abi.hip — engine + C ABI (include/unetpp.h) of the MI355X-native U-Net inference paths.
//
// arch 0, NestedUNet (reference src/models/unetpp.py:104-119, eval mode):
//   x0_0 = CB(3,32)(x)            x1_0 = CB(32,64)(pool x0_0)     x2_0 = CB(64,128)(pool x1_0)
//   x3_0 = CB(128,256)(pool x2_0) x4_0 = CB(256,512)(pool x3_0)
//   x3_1 = CB(768,256)(cat[x3_0, up x4_0])   x2_2 = CB(384,128)(cat[x2_0, up x3_1])
//   x1_3 = CB(192,64)(cat[x1_0, up x2_2])    x0_4 = CB(96,32)(cat[x0_0, up x1_3])
//   out  = Conv1x1(32,C)(x0_4)  -> argmax / class masks (infer_two_stage_burr.py:299-304)
// arch 1, SimpleUNet (reference src/models/simple_unet.py:94-128):
//   enc1 = CR(3,64) CR(64,64)(x)        enc2 = CR CR(pool enc1) [128]   enc3 [256]   enc4 [512]
//   dec3 = CR CR(cat[ConvT(512,256)(enc4), enc3])   dec2 = CR CR(cat[ConvT(256,128)(dec3), enc2])
//   dec1 = CR CR(cat[ConvT(128,64)(dec2), enc1])    out = Conv1x1(64,C)(dec1)
// Both are executed from one op list (convert, conv3x3, upsample, convT, head) built in unetpp_create.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <set>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/unetpp.h"
#include "aux_kernels.h"
#include "conv3x3_mfma.h"
#include "conv3x3_ws.h"
#include "convt2x2_mfma.h"
#include "tapmm_ws.h"

using namespace unetpp;

namespace {

thread_local std::string g_create_error;

constexpr uint32_t BLOB_MAGIC = 0x50504e55u;  // 'UNPP'
constexpr int BLOB_VERSION = 1;
const int NB[5] = {32, 64, 128, 256, 512};    // nb_filter, reference unetpp.py:49
const int SB[4] = {64, 128, 256, 512};        // SimpleUNet widths, simple_unet.py:32-57

struct Tensor {
  std::string name;
  size_t off = 0;   // byte offset inside one activation slot
  int C = 0;        // channels of an activation tensor; 0 = raw bytes (fp32 side tensors, split-K buffers)
  int lvl = 0;
  bool virt = false;   // named in the graph but never written (fused away): no storage
};

struct ConvLayer {
  std::string name;
  int cin_real = 0;   // channels the canonical weight has
  int cout = 0;
  int lvl = 0;
  int in = -1, in2 = -1, out = -1, pool = -1;   // tensor ids; in2: second source of the virtual concat
  bool do_pool = false;
  int zt = -1;                  // tensor id of the fp32 accumulator start values (tapmm_ws.h): K = skip channels only
  int y_t = -1, low_t = -1;     // tapmm: fp32 Y tensor and the low-res input it is computed from
  half_t* tapw = nullptr;       // tapmm: packed weights of the up channels
  bool c0f = false;             // conv0_0.conv2 computing conv0_0.conv1 itself from the caller's input (conv3x3_ws.h, C0F)
  bool upf = false;             // in2 is the LOW-resolution tensor: the loader does the bilinear x2 itself (no `up` tensor)
  size_t w_off = 0, b_off = 0;  // float offsets inside the canonical blob payload
  int KC = 16, NW = 1, MW = 2, WAVES = 8;
  int nchunks = 1;
  half_t* wpk = nullptr;
  float* scale = nullptr;
  float* mult = nullptr;
};

struct ConvTLayer {
  std::string name;
  int cin = 0, cout = 0, lvl = 0;   // lvl = level of the INPUT (the output is one level up)
  int in = -1, out = -1;
  size_t w_off = 0, b_off = 0;
  half_t* wpk = nullptr;
  float* scale = nullptr;
  float* mult = nullptr;
};

enum OpKind { OP_CONVERT, OP_CONV, OP_UP, OP_CONVT, OP_HEAD, OP_TAPMM, OP_UPSUM };
struct Op {
  OpKind kind;
  int idx = -1;            // conv / convT index, or (OP_UP) source tensor id
  int out = -1;            // OP_UP: destination tensor id; OP_HEAD: input tensor id
  bool fuse_head = false;  // OP_CONV: run the 1x1 head in this conv's epilogue when allowed
};

struct ProfRec {
  std::string name;
  double flops = 0, bytes = 0;
  int ev0 = -1, ev1 = -1;   // indices into the engine's event pool: launches on one stream are back to
                            // back, so a launch's start event is the previous launch's end event
};

}  // namespace

struct unetpp_engine {
  unetpp_config cfg{};
  int P = 2;
  bool x8 = false;                // UNETPP_PREC_EXACT8: P = 2 records with 8-bit cross-term planes (conv3x3_ws.h)
  int mb = 1;
  int num_cus = 256;
  std::string err;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  float* blob = nullptr;  // canonical fp32 blob payload on device (weights + biases)
  size_t blob_floats = 0;
  bool weights_loaded = false;
  std::vector<Tensor> tensors;
  std::vector<ConvLayer> convs;
  std::vector<ConvTLayer> convts;
  std::vector<Op> ops;
  int n_blob_layers = 0;          // header[4] of a matching blob
  int t_in8 = -1, t_head_in = -1;
  int head_cx = 32;
  size_t head_w_off = 0, head_b_off = 0;
  // profiling
  bool prof_on = false;
  std::vector<ProfRec> prof;
  int prof_used = 0;
  std::vector<hipEvent_t> evpool;
  int ev_used = 0;
  int prof_prev_ev = -1;          // end event of the previous launch on the same stream, -1 = none
  hipStream_t prof_prev_stream = nullptr;
  int last_b = 0, last_h = 0, last_w = 0;
  bool keep_all = false;   // debug: materialise the head's input and run the head as its own kernel
  // concurrent micro-batches: `nstreams` copies of the activation area, one internal stream each
  int nstreams = 1;
  size_t act_bytes = 0;          // size of one activation area (slot)
  size_t last_slot_off = 0;
  hipStream_t streams[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_start = nullptr, ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
  // frame glue: per-axis resize tables on the device, keyed by (kind, n_src, n_dst); kind 0 = linear, 1 = nearest
  std::map<std::tuple<int, int, int>, void*> resize_tabs;
  half_t* c1w = nullptr;          // fused first block: conv0_0.conv1 as MFMA A fragments (conv0_pack_kernel)
  int c0f_conv1 = -1;             // index of conv0_0.conv1 in `convs` when the first block is fused, else -1
  int ws_max_cout = 512;          // largest Cout the 8-row wave-specialised tiles are used for (UNETPP_WS_MAX_COUT; measured: every layer gains 1-8 %)
  bool ws64 = true;               // ... and for the Cout = 64 layers (UNETPP_NO_WS64=1: the lock-step kernel there)
  bool ws_cat = false;            // ... and for convs over two full-resolution sources (exact8; UNETPP_WS_CAT=1: in exact too)
  bool use_ws = true;             // exact-mode convs in the wave-specialised kernel (UNETPP_NO_WS=1: the lock-step one)
  unsigned* d_status = nullptr;   // sticky range flags (UNETPP_STATUS_*), one word inside the arena
  int ksplit_max = 16, ksplit_min_chunks = 4, ksplit_gate = 4;      // split-K of small launches (UNETPP_KSPLIT=max[,min chunks]; 1 = off)
  int t_kpart = -1, t_kcnt = -1;  // per slot: partial sums and arrival counters of the split tiles
  bool pair9 = true;              // EXACT8: ninth taps of consecutive chunks share an MFMA (UNETPP_NO_PAIR9=1: off; read at create)
  bool kcnt_dirty = false;        // a forward returned early: its split launches may have left counters behind
};

namespace {

int fail(unetpp_engine* e, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (e) e->err = buf; else g_create_error = buf;
  return code;
}

#define HIP_TRY(e, call)                                                                       \
  do {                                                                                         \
    hipError_t _s = (call);                                                                    \
    if (_s != hipSuccess) return fail(e, UNETPP_E_HIP, "%s: %s", #call, hipGetErrorString(_s)); \
  } while (0)

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Every entry point runs with the engine's device current and puts the caller's device back on return, so that a
// single-process multi-GPU program (torch, another engine) is not redirected by a call into this library.
struct DeviceScope {
  int prev = -1;
  bool switched = false;
  hipError_t st = hipSuccess;
  explicit DeviceScope(int dev) {
    st = hipGetDevice(&prev);
    if (st == hipSuccess && prev != dev) { st = hipSetDevice(dev); switched = st == hipSuccess; }
  }
  ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};
#define ENTER_DEVICE(e)                          \
  DeviceScope _dev_scope((e)->cfg.device);       \
  if (_dev_scope.st != hipSuccess) return fail(e, UNETPP_E_HIP, "hipSetDevice(%d): %s", (e)->cfg.device, hipGetErrorString(_dev_scope.st))

// payload floats of the canonical blob and number of layers in it
size_t blob_payload_floats(int arch, int C, int cin, int* n_layers = nullptr) {
  size_t n = 0;
  int layers = 0;
  if (arch == UNETPP_ARCH_NESTED) {
    int ci[9] = {cin, NB[0], NB[1], NB[2], NB[3], NB[3] + NB[4], NB[2] + NB[3], NB[1] + NB[2], NB[0] + NB[1]};
    int co[9] = {NB[0], NB[1], NB[2], NB[3], NB[4], NB[3], NB[2], NB[1], NB[0]};
    for (int b = 0; b < 9; ++b) {
      n += (size_t)co[b] * ci[b] * 9 + co[b];
      n += (size_t)co[b] * co[b] * 9 + co[b];
      layers += 2;
    }
    n += (size_t)C * NB[0] + C;
    layers += 1;
  } else {
    for (int l = 0; l < 4; ++l) {          // enc1..enc4
      int ci = l == 0 ? cin : SB[l - 1];
      n += (size_t)SB[l] * ci * 9 + SB[l] + (size_t)SB[l] * SB[l] * 9 + SB[l];
      layers += 2;
    }
    for (int l = 2; l >= 0; --l) { n += (size_t)SB[l + 1] * SB[l] * 4 + SB[l]; layers += 1; }   // up3, up2, up1
    for (int l = 2; l >= 0; --l) {         // dec3, dec2, dec1
      n += (size_t)SB[l] * (2 * SB[l]) * 9 + SB[l] + (size_t)SB[l] * SB[l] * 9 + SB[l];
      layers += 2;
    }
    n += (size_t)C * SB[0] + C;
    layers += 1;
  }
  if (n_layers) *n_layers = layers;
  return n;
}

// ---- conv dispatch ---------------------------------------------------------------------------
struct LaunchCtx { int device; int num_cus; int ksplit_max = 1, ksplit_min_chunks = 4, ksplit_gate = 4; };   // per engine: one process may drive engines on several devices

// The conv kernels take more dynamic LDS than the 64 KiB default: raise the function's limit to the whole 160 KiB
// once per (device, kernel).  The attribute is process-wide state of the HIP runtime and engines may be driven
// from several threads, so the bookkeeping is locked and the value is the same constant for everybody.
hipError_t allow_full_lds(const void* kernel, int device) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  std::lock_guard<std::mutex> lock(mu);
  if (done.count({device, kernel})) return hipSuccess;
  hipError_t st = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (st == hipSuccess) done.insert({device, kernel});
  return st;
}

template <int P, int KC, int NW, int MW, int WAVES, bool POOL, bool HEAD, bool UPF = false, bool ZINIT = false>
hipError_t launch_conv_k(const LaunchCtx& cx, const ConvArgs& a, hipStream_t s) {
  using C = ConvCfg<P, KC, NW, MW, WAVES, conv_single_stage<P, HEAD>(), UPF>;
  const int lds = C::LDS_BYTES + a.Cout * 8 + (HEAD ? ((a.head_C * 33 * 4 + 15) / 16) * 16 : 0);
  // persistent workgroups: as many as are resident at once, each walks tiles blockIdx, +grid, ...
  const int total = a.N * a.tiles_x * a.tiles_y * a.nct;
  const int per_cu = std::max(1, std::min(2, (160 * 1024) / lds));
  dim3 grid((unsigned)std::min(total, cx.num_cus * per_cu));
  auto k = conv3x3_bias_relu_kernel<P, KC, NW, MW, WAVES, POOL, HEAD, UPF, ZINIT>;
  hipError_t st = allow_full_lds((const void*)k, cx.device);
  if (st != hipSuccess) return st;
  hipLaunchKernelGGL(k, grid, dim3(C::NT), lds, s, a);
  return hipGetLastError();
}

// wave-specialised kernel (conv3x3_ws.h): 16-row tiles (Cout = 32) or 8-row tiles (Cout % 64 == 0), one persistent workgroup per CU
template <int P, bool POOL, bool HEAD, bool UPF, bool C0F = false, int NW = 1, int MW = 4, bool X8 = false, bool CAT2 = false>
hipError_t launch_ws_k(const LaunchCtx& cx, ConvArgs a, hipStream_t s) {
  using C = WsCfg<P, UPF, C0F, NW, MW, X8>;
  a.tiles_x = (a.W + C::TW - 1) / C::TW; a.tiles_y = (a.H + C::TH - 1) / C::TH; a.nct = a.Cout / C::BN;
  const int lds = C::LDS_BYTES + a.Cout * 8 + (HEAD ? ((a.head_C * 33 * 4 + 15) / 16) * 16 : 0);
  // Split-K plan (small batches): a layer with fewer tiles than a quarter of the CUs shares each tile's K-chunks among
  // `ksplit` workgroups -- the largest divisor of the chunk count that still leaves every workgroup four chunks and fits
  // the chip (measured at batch 1, 512x512: levels 3-4 gain 10-35 us per layer; two-way splits and two-chunk shares gain nothing:
  // a split launch costs its partials' round trip, ~10 us).
  // The plan depends on (batch, H, W) only; results of different plans differ in summation order (1e-7-class).
  const int tiles = a.N * a.tiles_x * a.tiles_y * a.nct;
  int ks = 1;
  if (!UPF && !C0F && !HEAD && a.kpart && a.kcnt && cx.ksplit_max > 1 && tiles * cx.ksplit_gate <= cx.num_cus)
    for (int d = 2; d <= cx.ksplit_max && tiles * d <= cx.num_cus && a.nchunks / d >= cx.ksplit_min_chunks; ++d)
      if (a.nchunks % d == 0 && !(a.pair9 && (a.nchunks / d) % 2)) ks = d;      // (pair9: whole chunk pairs per workgroup)
  a.ksplit = ks;
  const int total = tiles * ks;
  dim3 grid((unsigned)std::min(total, cx.num_cus));
  {      // the grid size in the kernel's tile-number radix (its workgroups step through the tiles by G)
    int t = (int)grid.x;
    a.gdec[0] = t % ks; t /= ks;
    a.gdec[1] = t % a.nct; t /= a.nct;
    a.gdec[2] = t % a.tiles_x; t /= a.tiles_x;
    a.gdec[3] = t % a.tiles_y;
    a.gdec[4] = t / a.tiles_y;
  }
  auto k = conv3x3_ws_kernel<P, POOL, HEAD, UPF, C0F, NW, MW, X8, CAT2>;
  hipError_t st = allow_full_lds((const void*)k, cx.device);
  if (st != hipSuccess) return st;
  hipLaunchKernelGGL(k, grid, dim3(C::NT), lds, s, a);
  return hipGetLastError();
}

template <bool X8>
hipError_t launch_ws_x(const LaunchCtx& cx, int P, const ConvArgs& a, bool pool, bool head, bool upf, bool c0f, hipStream_t s) {
  if (P != 2 || (upf && (pool || head)) || (pool && head)) return hipErrorInvalidValue;
  if (a.Cout >= 64 && a.Cout % 64 == 0) {      // 8-row tiles, two 32-channel blocks per consumer wave, Cout / 64 channel tiles
    if (head || c0f) return hipErrorInvalidValue;
    if (upf) return launch_ws_k<2, false, false, true, false, 2, 2, X8>(cx, a, s);
    if (pool) return a.in1 ? hipErrorInvalidValue : launch_ws_k<2, true, false, false, false, 2, 2, X8>(cx, a, s);
    if (a.in1) return launch_ws_k<2, false, false, false, false, 2, 2, X8, true>(cx, a, s);      // two full-resolution sources
    return launch_ws_k<2, false, false, false, false, 2, 2, X8>(cx, a, s);
  }
  if (a.in1 && !upf) return hipErrorInvalidValue;
  if (a.Cout != 32) return hipErrorInvalidValue;
  if (c0f) return (pool && !head && !upf && a.nchunks == 2) ? launch_ws_k<2, true, false, false, true, 1, 4, X8>(cx, a, s) : hipErrorInvalidValue;
  if (upf) return launch_ws_k<2, false, false, true, false, 1, 4, X8>(cx, a, s);
  if (head) return launch_ws_k<2, false, true, false, false, 1, 4, X8>(cx, a, s);
  if (pool) return launch_ws_k<2, true, false, false, false, 1, 4, X8>(cx, a, s);
  return launch_ws_k<2, false, false, false, false, 1, 4, X8>(cx, a, s);
}

hipError_t launch_ws(const LaunchCtx& cx, int P, bool x8, const ConvArgs& a, bool pool, bool head, bool upf, bool c0f, hipStream_t s) {
  return x8 ? launch_ws_x<true>(cx, P, a, pool, head, upf, c0f, s) : launch_ws_x<false>(cx, P, a, pool, head, upf, c0f, s);
}

template <int P, int KC, int NW, int MW, int WAVES>
hipError_t launch_conv_cfg(const LaunchCtx& cx, const ConvArgs& a, bool pool, bool head, bool upf, hipStream_t s) {
  if constexpr (NW == 1 && MW == 2 && KC == 16) {      // fused upsample: built for the narrow full-resolution tiles
    if (upf) return (pool || head) ? hipErrorInvalidValue : launch_conv_k<P, KC, NW, MW, WAVES, false, false, true>(cx, a, s);
  }
  if (upf) return hipErrorInvalidValue;
  if constexpr (P == 2 && NW == 2 && MW == 2 && KC == 16) {     // accumulators start from the low-resolution half (tapmm_ws.h)
    if (a.zinit) return (pool || head) ? hipErrorInvalidValue : launch_conv_k<P, KC, NW, MW, WAVES, false, false, false, true>(cx, a, s);
  }
  if (a.zinit) return hipErrorInvalidValue;
  if constexpr (NW == 1) {
    if (head) return launch_conv_k<P, KC, NW, MW, WAVES, false, true>(cx, a, s);
  }
  if constexpr (MW == 2) {     // the fused 2x2 pool needs both rows of a window in one wave
    if (pool) return launch_conv_k<P, KC, NW, MW, WAVES, true, false>(cx, a, s);
  }
  if (pool) return hipErrorInvalidValue;
  return launch_conv_k<P, KC, NW, MW, WAVES, false, false>(cx, a, s);
}

// `mw` = rows per wave: the layer's own (2: 16-row tiles) or 1 (8-row tiles, see small_grid_rows)
hipError_t launch_conv(const LaunchCtx& cx, int P, const ConvLayer& L, int mw, const ConvArgs& a, bool head, hipStream_t s) {
#define CASE(p, kc, nw, mw_, wv) \
  if (P == p && L.KC == kc && L.NW == nw && mw == mw_ && L.WAVES == wv) return launch_conv_cfg<p, kc, nw, mw_, wv>(cx, a, L.do_pool, head, L.upf, s);
  CASE(1, 16, 1, 2, 8)
  CASE(1, 32, 2, 2, 8)
  CASE(1, 16, 2, 2, 8)
  CASE(1, 16, 4, 2, 8)
  CASE(2, 16, 1, 2, 8)
  CASE(2, 16, 2, 2, 8)
  CASE(1, 16, 2, 1, 8)
  CASE(1, 32, 2, 1, 8)
  CASE(1, 16, 4, 1, 8)
  CASE(2, 16, 2, 1, 8)
#undef CASE
  return hipErrorInvalidValue;
}

// Small batches leave the deep layers with fewer 16x32-pixel tiles than the chip has CUs (B=1: 16 tiles at 32x32).
// Layers without a fused pool (its 2x2 window needs both rows in one wave) then run 8-row tiles: twice the
// workgroups, each with half the matrix work per K-chunk.  Every output is still accumulated chunk by chunk, tap by
// tap in the same order, so the result is bitwise the same whichever tile height ran (tested).
int small_grid_rows(const ConvLayer& L, int num_cus, int nb, int H, int W, bool head) {
  if (L.do_pool || head || L.upf || L.zt >= 0 || L.NW == 1 || L.MW != 2) return L.MW;
  const int tiles = nb * ((W + 31) / 32) * ((H + 15) / 16) * (L.cout / (32 * L.NW));
  return tiles < num_cus ? 1 : L.MW;
}

void choose_cfg(int P, int cin_tensor, ConvLayer& L) {
  L.MW = 2; L.WAVES = 8;
  if (P == 1) {
    if (L.cout == 32) { L.NW = 1; L.KC = 16; }     // 2 workgroups of 8 waves per CU (58 KB LDS each): measured best
    else if (L.cout == 64) { L.NW = 2; L.KC = (cin_tensor <= 16) ? 16 : 32; }
    else { L.NW = 4; L.KC = 16; }
  } else {
    L.KC = 16;
    L.NW = (L.cout == 32) ? 1 : 2;
  }
}

struct Launcher {
  unetpp_engine* e;
  hipStream_t s;
  int rc = UNETPP_OK;
  template <class F>
  void run(const std::string& name, double flops, double bytes, F&& f) {
    if (rc) return;
    ProfRec* r = nullptr;
    auto new_event = [&]() {
      if ((size_t)e->ev_used >= e->evpool.size()) {
        hipEvent_t ev = nullptr;
        hipError_t es = hipEventCreate(&ev);
        if (es != hipSuccess) { rc = fail(e, UNETPP_E_HIP, "hipEventCreate: %s", hipGetErrorString(es)); return -1; }
        e->evpool.push_back(ev);
      }
      return e->ev_used++;
    };
    if (e->prof_on) {
      if ((size_t)e->prof_used >= e->prof.size()) e->prof.push_back(ProfRec());
      r = &e->prof[e->prof_used++];
      r->name = name; r->flops = flops; r->bytes = bytes;
      if (e->prof_prev_ev >= 0 && e->prof_prev_stream == s) {
        r->ev0 = e->prof_prev_ev;
      } else {
        r->ev0 = new_event();
        if (r->ev0 < 0) { --e->prof_used; return; }
        (void)hipEventRecord(e->evpool[r->ev0], s);
      }
    }
    hipError_t st = f();
    if (st == hipSuccess) st = hipGetLastError();
    if (r) {
      r->ev1 = new_event();
      if (r->ev1 < 0) { --e->prof_used; e->prof_prev_ev = -1; return; }
      (void)hipEventRecord(e->evpool[r->ev1], s);
      e->prof_prev_ev = r->ev1; e->prof_prev_stream = s;
    }
    if (st != hipSuccess) rc = fail(e, UNETPP_E_HIP, "launch %s: %s", name.c_str(), hipGetErrorString(st));
  }
};

// ---- graph construction ------------------------------------------------------------------------
struct Builder {
  unetpp_engine* e;
  size_t act = 0;     // bytes of one activation slot so far
  size_t off = 0;     // floats of the blob payload so far
  int tensor(const std::string& name, int C, int lvl, bool virt = false) {
    Tensor t;
    t.name = name; t.C = C; t.lvl = lvl; t.off = act; t.virt = virt;
    const unetpp_config& c = e->cfg;
    size_t px = (size_t)e->mb * (c.max_h >> lvl) * (c.max_w >> lvl);
    if (!virt) act += align_up(px * e->P * C * sizeof(half_t), 256);
    e->tensors.push_back(t);
    return (int)e->tensors.size() - 1;
  }
  int tensor_raw(const std::string& name, size_t bytes_per_px, int lvl) {      // e.g. an fp32 tensor
    Tensor t;
    t.name = name; t.C = 0; t.lvl = lvl; t.off = act;
    const unetpp_config& c = e->cfg;
    size_t px = (size_t)e->mb * (c.max_h >> lvl) * (c.max_w >> lvl);
    act += align_up(px * bytes_per_px, 256);
    e->tensors.push_back(t);
    return (int)e->tensors.size() - 1;
  }
  int conv(const std::string& name, int cin_real, int in, int in2, int out, bool pool_to = false, int pool = -1, bool emit_op = true) {
    ConvLayer L;
    L.name = name; L.cin_real = cin_real; L.in = in; L.in2 = in2; L.out = out; L.pool = pool; L.do_pool = pool_to;
    L.cout = e->tensors[out].C; L.lvl = e->tensors[out].lvl;
    L.w_off = off; off += (size_t)L.cout * cin_real * 9;
    L.b_off = off; off += L.cout;
    const int c0 = e->tensors[in].C, c1 = in2 >= 0 ? e->tensors[in2].C : 0;
    choose_cfg(e->P, c0 + c1, L);
    L.nchunks = (c0 + L.KC - 1) / L.KC + c1 / L.KC;
    e->convs.push_back(L);
    Op op; op.kind = OP_CONV; op.idx = (int)e->convs.size() - 1;
    if (emit_op) e->ops.push_back(op);
    return op.idx;
  }
  void convt(const std::string& name, int in, int out) {
    ConvTLayer T;
    T.name = name; T.in = in; T.out = out; T.cin = e->tensors[in].C; T.cout = e->tensors[out].C; T.lvl = e->tensors[in].lvl;
    T.w_off = off; off += (size_t)T.cin * T.cout * 4;
    T.b_off = off; off += T.cout;
    e->convts.push_back(T);
    Op op; op.kind = OP_CONVT; op.idx = (int)e->convts.size() - 1;
    e->ops.push_back(op);
  }
  void up(int src, int dst) { Op op; op.kind = OP_UP; op.idx = src; op.out = dst; e->ops.push_back(op); }
  void head(int in, int cx) {
    e->t_head_in = in; e->head_cx = cx;
    e->head_w_off = off; off += (size_t)e->cfg.num_classes * cx;
    e->head_b_off = off; off += e->cfg.num_classes;
    Op op; op.kind = OP_HEAD; op.out = in; e->ops.push_back(op);
  }
};

void build_nested(unetpp_engine* e, Builder& b) {
  // exact mode: the first ConvBlock runs as ONE launch from the caller's tensor (no input copy, no conv0_0.conv1 launch,
  // no x0_0a tensor): conv3x3_ws.h, C0F.  UNETPP_NO_C0F=1 keeps the three launches (A/B measurements).
  const bool c0f = e->P == 2 && e->use_ws && !getenv("UNETPP_NO_C0F");
  if (!c0f) { Op cv; cv.kind = OP_CONVERT; e->ops.push_back(cv); }
  e->t_in8 = b.tensor("in8", 8, 0, c0f);            // fused first block: neither the fp16 input copy nor x0_0a exists
  int x[5], xa[5], pooled[4], up[4], d[4], da[4];
  for (int l = 0; l < 5; ++l) {
    char nm[32];
    snprintf(nm, sizeof nm, "x%d_0", l);
    xa[l] = b.tensor(std::string(nm) + "a", NB[l], l, c0f && l == 0);
    x[l] = b.tensor(nm, NB[l], l);
    if (l < 4) pooled[l] = b.tensor(std::string(nm) + "p", NB[l], l + 1);
    snprintf(nm, sizeof nm, "conv%d_0", l);
    const int c1 = b.conv(std::string(nm) + ".conv1", l == 0 ? 3 : NB[l - 1], l == 0 ? e->t_in8 : pooled[l - 1], -1, xa[l], false, -1,
                          !(c0f && l == 0));
    const int c2 = b.conv(std::string(nm) + ".conv2", NB[l], xa[l], -1, x[l], l < 4, l < 4 ? pooled[l] : -1);
    if (c0f && l == 0) { e->convs[c2].c0f = true; e->c0f_conv1 = c1; }
  }
  for (int l = 3; l >= 0; --l) {
    char nm[32], tn[32];
    snprintf(tn, sizeof tn, "x%d_%d", l, 4 - l);
    const int low = l == 3 ? x[4] : d[l + 1];
    // Level 0 (Cout = 32: narrow tiles, HBM co-bound): the decoder conv interpolates its `up` channels itself from
    // the low-res tensor (conv3x3_mfma.h, UPF) -- no upsample launch, no `up` tensor.  UNETPP_NO_UPF=1 keeps the
    // separate kernel (A/B measurements).
    // (EXACT8: every level that does not take the low-resolution GEMM below interpolates in its loader -- the separate
    // upsample kernel and the two-source loader do not know the 8-bit planes)
    const bool upf = e->x8 || (!getenv("UNETPP_NO_UPF") && (l == 0 || (l == 1 && e->P == 2 && e->ws64 && NB[1] <= e->ws_max_cout && !getenv("UNETPP_NO_UPF1"))));      // level 1: only the wave-specialised kernel has the fused loader for Cout = 64 (layer_uses_ws)
    // Levels 2-3 (exact mode): the up channels are multiplied at LOW resolution and interpolated afterwards
    // (tapmm_ws.h: half the flops of the layer); UNETPP_TAPMM=levels overrides, e.g. "" (off) or "123".
    const char* tl = getenv("UNETPP_TAPMM");
    // (the GEMM's 128-wide virtual-channel tiles need 9 * Cout % 128 == 0: levels 2 and 3; at level 1 the fp32
    // side tensors would be 0.9 GB per step and the path measured slower anyway)
    const bool tapmm = e->P == 2 && l >= 1 && (9 * NB[l]) % TapmmCfg::TN == 0 && strchr(tl ? tl : "23", '0' + l) != nullptr;
    if (tapmm) {
      snprintf(nm, sizeof nm, "conv%d_%d", l, 4 - l);
      const int yt = b.tensor_raw(std::string(tn) + "y", (size_t)9 * NB[l] * 4, l + 1);
      const int zt = b.tensor_raw(std::string(tn) + "z", (size_t)NB[l] * 4, l);
      da[l] = b.tensor(std::string(tn) + "a", NB[l], l);
      d[l] = b.tensor(tn, NB[l], l);
      { Op op; op.kind = OP_TAPMM; op.idx = (int)e->convs.size(); e->ops.push_back(op); }     // the conv pushed next
      { Op op; op.kind = OP_UPSUM; op.idx = (int)e->convs.size(); e->ops.push_back(op); }
      const int ci = b.conv(std::string(nm) + ".conv1", NB[l] + NB[l + 1], x[l], -1, da[l]);
      ConvLayer& L = e->convs[ci];
      L.zt = zt; L.y_t = yt; L.low_t = low;
      L.nchunks = NB[l] / L.KC;                         // K over the skip channels only
      b.conv(std::string(nm) + ".conv2", NB[l], da[l], -1, d[l]);
      continue;
    }
    if (!upf) up[l] = b.tensor(std::string(tn) + "u", NB[l + 1], l);
    da[l] = b.tensor(std::string(tn) + "a", NB[l], l);
    d[l] = b.tensor(tn, NB[l], l);
    if (!upf) b.up(low, up[l]);
    snprintf(nm, sizeof nm, "conv%d_%d", l, 4 - l);
    const int ci = b.conv(std::string(nm) + ".conv1", NB[l] + NB[l + 1], x[l], upf ? low : up[l], da[l]);     // cat([skip, up]) (unetpp.py:112-116)
    e->convs[ci].upf = upf;
    b.conv(std::string(nm) + ".conv2", NB[l], da[l], -1, d[l]);
    if (l == 0) e->ops.back().fuse_head = true;
  }
  b.head(d[0], NB[0]);
}

void build_simple(unetpp_engine* e, Builder& b) {
  Op cv; cv.kind = OP_CONVERT; e->ops.push_back(cv);
  e->t_in8 = b.tensor("in8", 8, 0);
  int enc[4], enca[4], pooled[3];
  for (int l = 0; l < 4; ++l) {
    char nm[32];
    snprintf(nm, sizeof nm, "enc%d", l + 1);
    enca[l] = b.tensor(std::string(nm) + "a", SB[l], l);
    enc[l] = b.tensor(nm, SB[l], l);
    if (l < 3) pooled[l] = b.tensor(std::string(nm) + "p", SB[l], l + 1);
    b.conv(std::string(nm) + ".0", l == 0 ? 3 : SB[l - 1], l == 0 ? e->t_in8 : pooled[l - 1], -1, enca[l]);
    b.conv(std::string(nm) + ".2", SB[l], enca[l], -1, enc[l], l < 3, l < 3 ? pooled[l] : -1);
  }
  // The canonical blob follows the module DEFINITION order (enc*, up3, up2, up1, dec3, dec2, dec1, final) while
  // the forward interleaves up/dec: blob offsets are assigned here in definition order, ops in forward order.
  size_t up_w[3], up_b[3], dec_w[3][2], dec_b[3][2];
  size_t off = b.off;
  for (int l = 2; l >= 0; --l) { up_w[l] = off; off += (size_t)SB[l + 1] * SB[l] * 4; up_b[l] = off; off += SB[l]; }
  for (int l = 2; l >= 0; --l) {
    dec_w[l][0] = off; off += (size_t)SB[l] * 2 * SB[l] * 9; dec_b[l][0] = off; off += SB[l];
    dec_w[l][1] = off; off += (size_t)SB[l] * SB[l] * 9; dec_b[l][1] = off; off += SB[l];
  }
  int prev = enc[3];
  for (int l = 2; l >= 0; --l) {
    char nm[32];
    snprintf(nm, sizeof nm, "up%d", l + 1);
    int u = b.tensor(std::string(nm) + "t", SB[l], l);
    b.convt(nm, prev, u);
    e->convts.back().w_off = up_w[l]; e->convts.back().b_off = up_b[l];
    snprintf(nm, sizeof nm, "dec%d", l + 1);
    int da = b.tensor(std::string(nm) + "a", SB[l], l);
    int d = b.tensor(nm, SB[l], l);
    int c1 = b.conv(std::string(nm) + ".0", 2 * SB[l], u, enc[l], da);        // cat([up, enc]) (simple_unet.py:112,117,122)
    e->convs[c1].w_off = dec_w[l][0]; e->convs[c1].b_off = dec_b[l][0];
    int c2 = b.conv(std::string(nm) + ".2", SB[l], da, -1, d);
    e->convs[c2].w_off = dec_w[l][1]; e->convs[c2].b_off = dec_b[l][1];
    prev = d;
  }
  b.off = off;
  b.head(prev, SB[0]);
}

}  // namespace

extern "C" {

#ifndef UNETPP_SRC_HASH
#define UNETPP_SRC_HASH "unknown"
#endif
// "src:<hash>" = digest of the sources this binary was built from (unet-_amd/_lib.py compares it with the tree)
#ifdef UNETPP_WS_DBG
#define UNETPP_BUILD_TAG " +wsdbg"      /* measurement build (phase ablations): unet-_amd/_lib.py refuses it unless asked */
#else
#define UNETPP_BUILD_TAG ""
#endif
const char* unetpp_version(void) { return "unetpp-hip 0.3.0 (gfx950) src:" UNETPP_SRC_HASH UNETPP_BUILD_TAG; }

const char* unetpp_last_error(const unetpp_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

size_t unetpp_weights_blob_bytes(int num_classes, int in_channels) {
  return 32 + 4 * blob_payload_floats(UNETPP_ARCH_NESTED, num_classes, in_channels);
}

size_t unetpp_weights_blob_bytes_arch(int arch, int num_classes, int in_channels) {
  if (arch != UNETPP_ARCH_NESTED && arch != UNETPP_ARCH_SIMPLE) return 0;
  return 32 + 4 * blob_payload_floats(arch, num_classes, in_channels);
}

int unetpp_create(const unetpp_config* cfg, unetpp_engine** out) {
  if (!cfg || !out) return fail(nullptr, UNETPP_E_INVALID, "null argument");
  *out = nullptr;
  if (cfg->arch != UNETPP_ARCH_NESTED && cfg->arch != UNETPP_ARCH_SIMPLE) return fail(nullptr, UNETPP_E_INVALID, "arch=%d unknown", cfg->arch);
  if (cfg->in_channels != 3) return fail(nullptr, UNETPP_E_UNSUPPORTED, "input_channels=%d unsupported (only 3)", cfg->in_channels);
  if (cfg->num_classes < 1 || cfg->num_classes > HEAD_FUSED_MAX_CLASSES)
    return fail(nullptr, UNETPP_E_INVALID, "num_classes=%d out of range [1,%d]", cfg->num_classes, HEAD_FUSED_MAX_CLASSES);
  const int mult = cfg->arch == UNETPP_ARCH_NESTED ? 16 : 8;      // 4 resp. 3 poolings
  if (cfg->max_batch < 1 || cfg->max_h < mult || cfg->max_w < mult || cfg->max_h % mult || cfg->max_w % mult)
    return fail(nullptr, UNETPP_E_INVALID, "max shape (%d,%d,%d): batch>=1 and H,W positive multiples of %d required",
                cfg->max_batch, cfg->max_h, cfg->max_w, mult);
  if (cfg->precision != UNETPP_PREC_EXACT && cfg->precision != UNETPP_PREC_FAST && cfg->precision != UNETPP_PREC_EXACT8)
    return fail(nullptr, UNETPP_E_INVALID, "precision=%d unknown", cfg->precision);
  if (cfg->precision == UNETPP_PREC_EXACT8) {
    // EXACT8 exists in the wave-specialised kernels only: no lock-step / unfused alternatives to switch to
    // (SimpleUNet: every conv through the wave-specialised kernel -- two full-resolution sources included --, the transposed
    // convs on fp16 terms decoded from the 8-bit residual plane, convt2x2_mfma.h)
    for (const char* sw : {"UNETPP_NO_WS", "UNETPP_NO_WS64", "UNETPP_WS_MAX_COUT", "UNETPP_NO_C0F", "UNETPP_NO_UPF", "UNETPP_NO_UPF1"})
      if (getenv(sw)) return fail(nullptr, UNETPP_E_UNSUPPORTED, "%s has no meaning with precision EXACT8", sw);
  }
  // the conv loader addresses one image of one tensor with 32-bit byte offsets (buffer loads): the largest
  // full-resolution tensor has 64 channels x (1 or 2) fp16 planes
  if ((double)cfg->max_h * cfg->max_w * 64 * 2 * (cfg->precision == UNETPP_PREC_FAST ? 1 : 2) >= 2147483648.0)
    return fail(nullptr, UNETPP_E_UNSUPPORTED, "max shape %dx%d too large for 32-bit tensor offsets", cfg->max_h, cfg->max_w);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, UNETPP_E_HIP, "no HIP device available: this engine has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, UNETPP_E_INVALID, "device %d not in [0,%d)", cfg->device, ndev);
  DeviceScope dev_scope(cfg->device);
  if (dev_scope.st != hipSuccess) return fail(nullptr, UNETPP_E_HIP, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(dev_scope.st));

  unetpp_engine* e = new unetpp_engine();
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) e->num_cus = prop.multiProcessorCount;
  }
  e->cfg = *cfg;
  e->use_ws = !getenv("UNETPP_NO_WS");
  e->ws64 = e->use_ws && !getenv("UNETPP_NO_WS64");
  if (const char* mc = getenv("UNETPP_WS_MAX_COUT")) e->ws_max_cout = atoi(mc);
  e->P = cfg->precision == UNETPP_PREC_FAST ? 1 : 2;
  e->x8 = cfg->precision == UNETPP_PREC_EXACT8;
  // convs over two full-resolution sources: the wave-specialised kernel in exact8 (the only kernel with that arithmetic);
  // exact keeps the lock-step kernel (same speed, and its results do not depend on a split-K plan) unless UNETPP_WS_CAT=1
  { const char* c = getenv("UNETPP_WS_CAT"); e->ws_cat = e->use_ws && (e->x8 || (c && c[0] == '1')); }
  e->mb = (cfg->micro_batch > 0 && cfg->micro_batch < cfg->max_batch) ? cfg->micro_batch : cfg->max_batch;
  e->nstreams = std::max(1, std::min(4, cfg->streams));
  if (e->mb >= cfg->max_batch) e->nstreams = 1;      // a single pass has nothing to overlap with
  const int P = e->P;

  e->pair9 = !getenv("UNETPP_NO_PAIR9");
  if (const char* k = getenv("UNETPP_KSPLIT")) {
    e->ksplit_max = std::max(1, atoi(k));
    if (const char* c = strchr(k, ',')) {
      e->ksplit_min_chunks = std::max(1, atoi(c + 1));
      if (const char* g2 = strchr(c + 1, ',')) e->ksplit_gate = std::max(1, atoi(g2 + 1));      // split when tiles * gate <= CUs
    }
  }
  Builder b{e};
  if (cfg->arch == UNETPP_ARCH_NESTED) build_nested(e, b); else build_simple(e, b);
  if (e->P == 2 && e->use_ws && e->ksplit_max > 1) {
    // at most num_cus workgroups take part in a split launch, each with 4 consumer waves x 16 KB of raw accumulators
    e->t_kpart = (int)e->tensors.size();
    { Tensor t; t.name = "ksplit.partials"; t.off = b.act; e->tensors.push_back(t); b.act += align_up((size_t)e->num_cus * 4 * 16384, 256); }
    e->t_kcnt = (int)e->tensors.size();
    { Tensor t; t.name = "ksplit.counters"; t.off = b.act; e->tensors.push_back(t); b.act += align_up((size_t)e->num_cus * 4 * sizeof(unsigned), 256); }
  }
  e->blob_floats = b.off;
  if (b.off != blob_payload_floats(cfg->arch, cfg->num_classes, 3, &e->n_blob_layers)) {
    delete e;
    return fail(nullptr, UNETPP_E_STATE, "internal: blob size mismatch");
  }
  e->act_bytes = align_up(b.act, 4096);
  size_t total = e->act_bytes * e->nstreams;

  // ---- blob + packed weights + scales in the same arena
  size_t blob_off = total; total += align_up(b.off * sizeof(float), 256);
  std::vector<size_t> wpk_off(e->convs.size()), sc_off(e->convs.size()), mu_off(e->convs.size());
  for (size_t i = 0; i < e->convs.size(); ++i) {
    ConvLayer& L = e->convs[i];
    size_t wb = (size_t)(L.cout / (32 * L.NW)) * L.nchunks * P * 9 * L.KC * (32 * L.NW) * sizeof(half_t);
    if (e->x8) wb = (size_t)(L.cout / (32 * L.NW)) * L.nchunks * 38 * (32 * L.NW) * 16;      // WsCfg::SLAB_BYTES with X8
    wpk_off[i] = total; total += align_up(wb, 256);
    sc_off[i] = total; total += align_up(L.cout * sizeof(float), 256);
    mu_off[i] = total; total += align_up(L.cout * sizeof(float), 256);
  }
  std::vector<size_t> tapw_off(e->convs.size(), 0);
  for (size_t i = 0; i < e->convs.size(); ++i) {
    ConvLayer& L = e->convs[i];
    if (L.zt < 0) continue;
    const int cup = L.cin_real - e->tensors[L.in].C;
    tapw_off[i] = total; total += align_up((size_t)9 * L.cout * cup * P * sizeof(half_t), 256);
  }
  std::vector<size_t> twpk(e->convts.size()), tsc(e->convts.size()), tmu(e->convts.size());
  for (size_t i = 0; i < e->convts.size(); ++i) {
    ConvTLayer& T = e->convts[i];
    twpk[i] = total; total += align_up((size_t)4 * T.cout * T.cin * P * sizeof(half_t), 256);
    tsc[i] = total; total += align_up(T.cout * sizeof(float), 256);
    tmu[i] = total; total += align_up(T.cout * sizeof(float), 256);
  }
  const size_t c1w_off = total; total += 4096;
  const size_t status_off = total; total += 256;
  hipError_t st = hipMalloc((void**)&e->arena, total);
  if (st == hipSuccess) {
    st = hipMemset(e->arena + status_off, 0, 256);
    if (st != hipSuccess) { (void)hipFree(e->arena); e->arena = nullptr; }
  }
  if (st != hipSuccess) {
    std::string m = hipGetErrorString(st);
    delete e;
    return fail(nullptr, UNETPP_E_HIP, "hipMalloc(%zu bytes): %s", total, m.c_str());
  }
  e->arena_bytes = total;
  if (e->t_kcnt >= 0)
    for (int i = 0; i < e->nstreams; ++i)
      (void)hipMemset(e->arena + (size_t)i * e->act_bytes + e->tensors[e->t_kcnt].off, 0, (size_t)e->num_cus * 4 * sizeof(unsigned));
  e->d_status = (unsigned*)(e->arena + status_off);
  e->c1w = (half_t*)(e->arena + c1w_off);
  e->blob = (float*)(e->arena + blob_off);
  for (size_t i = 0; i < e->convs.size(); ++i) {
    e->convs[i].wpk = (half_t*)(e->arena + wpk_off[i]);
    e->convs[i].scale = (float*)(e->arena + sc_off[i]);
    e->convs[i].mult = (float*)(e->arena + mu_off[i]);
    if (e->convs[i].zt >= 0) e->convs[i].tapw = (half_t*)(e->arena + tapw_off[i]);
  }
  for (size_t i = 0; i < e->convts.size(); ++i) {
    e->convts[i].wpk = (half_t*)(e->arena + twpk[i]);
    e->convts[i].scale = (float*)(e->arena + tsc[i]);
    e->convts[i].mult = (float*)(e->arena + tmu[i]);
  }
  if (e->nstreams > 1) {
    hipError_t se = hipSuccess;
    for (int i = 0; i < e->nstreams && se == hipSuccess; ++i) {
      se = hipStreamCreateWithFlags(&e->streams[i], hipStreamNonBlocking);
      if (se == hipSuccess) se = hipEventCreateWithFlags(&e->ev_done[i], hipEventDisableTiming);
    }
    if (se == hipSuccess) se = hipEventCreateWithFlags(&e->ev_start, hipEventDisableTiming);
    if (se != hipSuccess) {
      std::string m = hipGetErrorString(se);
      unetpp_destroy(e);                       // frees the arena and whatever streams/events exist
      return fail(nullptr, UNETPP_E_HIP, "stream/event creation: %s", m.c_str());
    }
  }
  *out = e;
  return UNETPP_OK;
}

void unetpp_destroy(unetpp_engine* e) {
  if (!e) return;
  DeviceScope dev_scope(e->cfg.device);
  for (auto ev : e->evpool) (void)hipEventDestroy(ev);
  for (int i = 0; i < 4; ++i) { if (e->streams[i]) (void)hipStreamDestroy(e->streams[i]); if (e->ev_done[i]) (void)hipEventDestroy(e->ev_done[i]); }
  if (e->ev_start) (void)hipEventDestroy(e->ev_start);
  if (e->arena) (void)hipFree(e->arena);
  for (auto& kv : e->resize_tabs) (void)hipFree(kv.second);
  delete e;
}

size_t unetpp_workspace_bytes(const unetpp_engine* e) { return e ? e->arena_bytes : 0; }

// which conv layers run in the wave-specialised kernel (a property of the engine and the layer, not of a call)
#ifdef UNETPP_WS_DBG
static unsigned long long* unetpp_dbg_stamp_buf = nullptr;     // measurement build: in-kernel stamps (scripts/ws_stamps.sh)
#endif
static bool layer_uses_ws(const unetpp_engine* e, const ConvLayer& L) {
  // two full-resolution sources (SimpleUNet's decoder conv1): whole 16-channel records in both, same kernel (exact8, or UNETPP_WS_CAT=1)
  const bool cat_ok = L.in2 >= 0 && !L.upf && e->ws_cat && e->tensors[L.in].C % 16 == 0 && e->tensors[L.in2].C % 16 == 0 && !L.do_pool && L.cout % 64 == 0;
  return e->use_ws && e->P == 2 && (L.cout == 32 || (e->ws64 && L.cout >= 64 && L.cout <= e->ws_max_cout)) && (L.in2 < 0 || L.upf || cat_ok);
}

// EXACT8: layers whose chunk count is even pair the ninth taps of consecutive chunks (conv3x3_ws.h); the split-K plan then
// only makes shares with an even number of chunks (launch_ws_k)
static bool x8_pair9(const unetpp_engine* e, const ConvLayer& L) { return e->x8 && e->pair9 && L.nchunks % 2 == 0; }

static int repack(unetpp_engine* e, hipStream_t s) {
  const int P = e->P;
  for (auto& L : e->convs) {
    const float* w = e->blob + L.w_off;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(L.cout), dim3(256), 0, s, w, L.cin_real * 9, e->blob + L.b_off, L.mult, L.scale, e->d_status);
    const int BN = 32 * L.NW;
    if (e->x8) {
      if ((int)(&L - &e->convs[0]) == e->c0f_conv1) continue;      // conv0_0.conv1 runs in the producers (conv0_pack_kernel below)
      const long long units = (long long)(L.cout / BN) * L.nchunks * 38 * BN;
      hipLaunchKernelGGL(weight_pack_x8_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, w, L.mult, L.cin_real, L.cout, BN,
                         L.nchunks, (char*)L.wpk, units, x8_pair9(e, L) ? 1 : 0);
      continue;
    }
    long long units = (long long)(L.cout / BN) * L.nchunks * P * 9 * (L.KC / 8) * BN;
    hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, w, L.mult, L.cin_real,
                       L.cout, P, L.KC, BN, L.nchunks, L.wpk, units, layer_uses_ws(e, L) ? 1 : 0);
  }
  for (auto& L : e->convs) {
    if (L.zt < 0) continue;
    const int cs = e->tensors[L.in].C, cup = L.cin_real - cs;
    const long long units = (long long)9 * L.cout * cup * P / 8;
    hipLaunchKernelGGL(tapw_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, e->blob + L.w_off, L.mult, L.cout, cs,
                       cup, L.tapw, units, e->x8 ? 1 : 0);
  }
  if (e->c0f_conv1 >= 0) {
    const ConvLayer& L1 = e->convs[e->c0f_conv1];
    hipLaunchKernelGGL(conv0_pack_kernel, dim3(1), dim3(256), 0, s, e->blob + L1.w_off, L1.mult, e->c1w);
  }
  for (auto& T : e->convts) {
    const float* w = e->blob + T.w_off;
    hipLaunchKernelGGL(convt_scale_kernel, dim3(T.cout), dim3(256), 0, s, w, T.cin, T.cout, e->blob + T.b_off, T.mult, T.scale, e->d_status);
    long long units = (long long)(4 * T.cout / 32) * (T.cin / 16) * P * 64;
    hipLaunchKernelGGL(convt_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, w, T.mult, T.cin, T.cout, P,
                       T.wpk, units);
  }
  HIP_TRY(e, hipGetLastError());
  e->weights_loaded = true;
  return UNETPP_OK;
}

static int check_header(unetpp_engine* e, const uint32_t* h, size_t bytes) {
  const size_t want = unetpp_weights_blob_bytes_arch(e->cfg.arch, e->cfg.num_classes, e->cfg.in_channels);
  if (bytes != want)
    return fail(e, UNETPP_E_INVALID, "weight blob is %zu bytes, expected %zu for arch=%d num_classes=%d", bytes, want, e->cfg.arch,
                e->cfg.num_classes);
  if (h[0] != BLOB_MAGIC || (int)h[1] != BLOB_VERSION) return fail(e, UNETPP_E_INVALID, "bad weight blob magic/version");
  if ((int)h[2] != e->cfg.num_classes || (int)h[3] != e->cfg.in_channels || (int)h[4] != e->n_blob_layers || (int)h[5] != e->cfg.arch)
    return fail(e, UNETPP_E_INVALID, "weight blob is for num_classes=%u in_channels=%u layers=%u arch=%u", h[2], h[3], h[4], h[5]);
  return UNETPP_OK;
}

int unetpp_load_weights(unetpp_engine* e, const void* host_blob, size_t bytes) {
  if (!e || !host_blob) return fail(e, UNETPP_E_INVALID, "null argument");
  ENTER_DEVICE(e);
  if (bytes < 32) return fail(e, UNETPP_E_INVALID, "weight blob too small");
  int rc = check_header(e, (const uint32_t*)host_blob, bytes);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpy(e->blob, (const char*)host_blob + 32, bytes - 32, hipMemcpyHostToDevice));
  rc = repack(e, nullptr);
  if (rc) return rc;
  HIP_TRY(e, hipDeviceSynchronize());
  return UNETPP_OK;
}

int unetpp_load_weights_device(unetpp_engine* e, const void* dev_blob, size_t bytes, void* stream) {
  if (!e || !dev_blob) return fail(e, UNETPP_E_INVALID, "null argument");
  ENTER_DEVICE(e);
  if (bytes < 32) return fail(e, UNETPP_E_INVALID, "weight blob too small");
  uint32_t h[8];
  hipStream_t s = (hipStream_t)stream;
  HIP_TRY(e, hipMemcpyAsync(h, dev_blob, 32, hipMemcpyDeviceToHost, s));
  HIP_TRY(e, hipStreamSynchronize(s));
  int rc = check_header(e, h, bytes);
  if (rc) return rc;
  HIP_TRY(e, hipMemcpyAsync(e->blob, (const char*)dev_blob + 32, bytes - 32, hipMemcpyDeviceToDevice, s));
  return repack(e, s);
}

// ---- forward -------------------------------------------------------------------------------------
int unetpp_forward(unetpp_engine* e, const void* dev_input, int in_format, int batch, int h, int w, float* dev_logits,
                   uint8_t* dev_mask, uint8_t* dev_cable, uint8_t* dev_tape, void* stream) {
  unetpp_outputs o{};
  o.dev_logits = dev_logits; o.dev_mask = dev_mask; o.dev_cable = dev_cable; o.dev_tape = dev_tape;
  o.rule = UNETPP_RULE_ARGMAX;
  return unetpp_forward_ex(e, dev_input, in_format, batch, h, w, &o, stream);
}

int unetpp_forward_ex(unetpp_engine* e, const void* dev_input, int in_format, int batch, int h, int w,
                      const unetpp_outputs* outp, void* stream) {
  if (!e) return UNETPP_E_INVALID;
  if (!outp) return fail(e, UNETPP_E_INVALID, "outputs is NULL");
  float* dev_logits = outp->dev_logits; float* dev_probs = outp->dev_probs;
  uint8_t* dev_mask = outp->dev_mask; uint8_t* dev_cable = outp->dev_cable; uint8_t* dev_tape = outp->dev_tape;
  if (outp->rule < UNETPP_RULE_ARGMAX || outp->rule > UNETPP_RULE_EXCLUSIVE) return fail(e, UNETPP_E_INVALID, "unknown rule %d", outp->rule);
  if (outp->rule != UNETPP_RULE_ARGMAX && e->cfg.num_classes < 3)
    return fail(e, UNETPP_E_INVALID, "class rules need num_classes >= 3 (bg, cable, tape)");
  if (!dev_input) return fail(e, UNETPP_E_INVALID, "input is NULL");
  if (!e->weights_loaded) return fail(e, UNETPP_E_STATE, "forward before load_weights");
  if (in_format != UNETPP_IN_F32_NCHW && in_format != UNETPP_IN_U8_NHWC_BGR) return fail(e, UNETPP_E_INVALID, "unknown input format %d", in_format);
  if (batch < 1 || batch > e->cfg.max_batch) return fail(e, UNETPP_E_INVALID, "batch %d not in [1,%d]", batch, e->cfg.max_batch);
  const int mult = e->cfg.arch == UNETPP_ARCH_NESTED ? 16 : 8;
  if (h < mult || w < mult || h % mult || w % mult)
    return fail(e, UNETPP_E_INVALID, "Sizes of tensors must match: H=%d W=%d must be positive multiples of %d", h, w, mult);
  if (h > e->cfg.max_h || w > e->cfg.max_w) return fail(e, UNETPP_E_INVALID, "shape %dx%d exceeds engine maximum %dx%d", h, w, e->cfg.max_h, e->cfg.max_w);
  ENTER_DEVICE(e);
  hipStream_t user_stream = (hipStream_t)stream;
  hipStream_t s = user_stream;
  const int P = e->P, C = e->cfg.num_classes;
  const bool multi = e->nstreams > 1 && batch > e->mb;
  // join: the caller's stream continues after every internal stream has drained.  It runs on EVERY exit after the
  // fork, failures included: the caller may free or reuse its output buffers on its stream right after an error
  // return, and kernels already queued on the internal streams still write to them.
  struct Join {
    unetpp_engine* e; hipStream_t user; bool armed = false;
    hipError_t run() {
      if (!armed) return hipSuccess;
      armed = false;
      hipError_t first = hipSuccess;
      for (int i = 0; i < e->nstreams; ++i) {
        hipError_t r = hipEventRecord(e->ev_done[i], e->streams[i]);
        if (r == hipSuccess) r = hipStreamWaitEvent(user, e->ev_done[i], 0);
        if (r != hipSuccess) { (void)hipStreamSynchronize(e->streams[i]); if (first == hipSuccess) first = r; }
      }
      return first;
    }
    ~Join() { (void)run(); }
  } join{e, user_stream};
  if (multi) {   // fork: the internal streams start after everything already queued on the caller's stream
    HIP_TRY(e, hipEventRecord(e->ev_start, user_stream));
    join.armed = true;
    for (int i = 0; i < e->nstreams; ++i) HIP_TRY(e, hipStreamWaitEvent(e->streams[i], e->ev_start, 0));
  }
  e->last_b = batch; e->last_h = h; e->last_w = w;
  // The last arriver of every split tile puts its counter back to zero, so a completed forward leaves them clean; after a
  // forward that failed half-way they are cleared here (stream-ordered, before the first launch).
  if (e->t_kcnt >= 0 && e->kcnt_dirty)
    for (int i = 0; i < e->nstreams; ++i)
      HIP_TRY(e, hipMemsetAsync(e->arena + (size_t)i * e->act_bytes + e->tensors[e->t_kcnt].off, 0, (size_t)e->num_cus * 4 * sizeof(unsigned),
                                multi ? e->streams[i] : user_stream));
  e->kcnt_dirty = true;
  Launcher Lx{e, s};
  int pass = 0;
  e->prof_prev_ev = -1;     // other work may sit between two forwards: start a fresh event chain
  const size_t hw = (size_t)h * w;

  for (int b0 = 0; b0 < batch; b0 += e->mb, ++pass) {
    const int nb = std::min(e->mb, batch - b0);
    const int slot = multi ? pass % e->nstreams : 0;
    if (multi) { s = e->streams[slot]; Lx.s = s; }
    const size_t slot_off = (size_t)slot * e->act_bytes;
    e->last_slot_off = slot_off;
    auto tp = [&](int id) { return (half_t*)(e->arena + slot_off + e->tensors[id].off); };
    float* lg = dev_logits ? dev_logits + (size_t)b0 * C * hw : nullptr;
    uint8_t* mk = dev_mask ? dev_mask + (size_t)b0 * hw : nullptr;
    uint8_t* cb = dev_cable ? dev_cable + (size_t)b0 * hw : nullptr;
    uint8_t* tpe = dev_tape ? dev_tape + (size_t)b0 * hw : nullptr;
    float* pr = dev_probs ? dev_probs + (size_t)b0 * C * hw : nullptr;
    bool head_done = false;
#ifdef UNETPP_WS_DBG
    std::vector<std::string> dbg_timeline;                   // UNETPP_WS_STAMPS=all: the wave-specialised launches of this pass, in order
#endif

    for (const Op& op : e->ops) {
      if (op.kind == OP_CONVERT) {
        const char* src = (const char*)dev_input + (in_format == UNETPP_IN_F32_NCHW ? (size_t)b0 * 3 * hw * 4 : (size_t)b0 * hw * 3);
        size_t total = (size_t)nb * hw;
        double bytes = (double)total * (in_format == UNETPP_IN_F32_NCHW ? 12 : 3) + (double)total * P * 16;
        Lx.run(e->x8 ? "convert_input|convert_input_kernel<2, true>" : P == 2 ? "convert_input|convert_input_kernel<2>" : "convert_input|convert_input_kernel<1>", 0, bytes, [&] {
          if (e->x8) hipLaunchKernelGGL((convert_input_kernel<2, true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const void*)src, in_format, nb, h, w, tp(e->t_in8), e->d_status);
          else if (P == 2) hipLaunchKernelGGL(convert_input_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const void*)src, in_format, nb, h, w, tp(e->t_in8), e->d_status);
          else hipLaunchKernelGGL(convert_input_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const void*)src, in_format, nb, h, w, tp(e->t_in8), e->d_status);
          return hipSuccess;
        });
      } else if (op.kind == OP_CONV) {
        ConvLayer& L = e->convs[op.idx];
        const bool head = op.fuse_head && !e->keep_all && L.cout == 32;
        ConvArgs a{};
        const int H = h >> L.lvl, W = w >> L.lvl;
        const Tensor& t0 = e->tensors[L.in];
        const int c1 = L.in2 >= 0 ? e->tensors[L.in2].C : 0;
        a.in0 = tp(L.in); a.in1 = L.in2 >= 0 ? tp(L.in2) : nullptr; a.C0 = t0.C; a.C1 = c1;
        a.wpk = L.wpk; a.scale = L.scale; a.bias = e->blob + L.b_off; a.out = tp(L.out);
        a.pool_out = L.do_pool ? tp(L.pool) : nullptr;
        a.N = nb; a.H = H; a.W = W; a.Cout = L.cout;
        a.status = e->d_status;
        a.zinit = L.zt >= 0 ? (const float*)tp(L.zt) : nullptr;
        if (e->t_kpart >= 0) { a.kpart = (float*)tp(e->t_kpart); a.kcnt = (unsigned*)tp(e->t_kcnt); }
        a.ksplit = 1;
        a.pair9 = x8_pair9(e, L) ? 1 : 0;
        if (L.c0f) {     // the first block reads the caller's tensor itself
          const ConvLayer& L1 = e->convs[e->c0f_conv1];
          a.raw_in = (const char*)dev_input + (in_format == UNETPP_IN_F32_NCHW ? (size_t)b0 * 3 * hw * 4 : (size_t)b0 * hw * 3);
          a.raw_fmt = in_format == UNETPP_IN_F32_NCHW ? 0 : 1;
          a.c1w = e->c1w; a.c1_scale = L1.scale; a.c1_bias = e->blob + L1.b_off;
        }
#ifdef UNETPP_WS_DBG
        { const char* d = getenv("UNETPP_WS_DBG"); a.dbg = d ? atoi(d) : 0; }
        // UNETPP_WS_DBG_ONLY=layer: the ablation applies to that launch alone -- its inputs stay genuine (a layer fed with the
        // zeros an ablated predecessor leaves behind runs at a different clock)
        { const char* only = getenv("UNETPP_WS_DBG_ONLY"); if (only && L.name != only) a.dbg = 0; }
        unsigned long long*& stamp_buf = unetpp_dbg_stamp_buf;
        const char* stamp_layer = getenv("UNETPP_WS_STAMPS");      // layer name: print that launch's in-kernel phase times
        const bool stamp_all = stamp_layer && !strcmp(stamp_layer, "all");      // "all": one wall-clock timeline of the forward
        const bool stamp_this = stamp_layer && L.name == stamp_layer;
        if (stamp_this || stamp_all) {
          if (!stamp_buf) { (void)hipMalloc((void**)&stamp_buf, (size_t)32 * 1024 * 32 * 8); (void)hipMemset(stamp_buf, 0, (size_t)32 * 1024 * 32 * 8); }
          if (stamp_this) (void)hipMemsetAsync(stamp_buf, 0, 1024 * 32 * 8 * 5, s);
          a.stamps = stamp_buf + (stamp_all ? (size_t)(1 + dbg_timeline.size()) * 1024 * 32 : 0);
          if (stamp_all) dbg_timeline.push_back(L.name);
        }
#endif
        const int mw = small_grid_rows(L, e->num_cus, nb, H, W, head);
        const int TH = L.WAVES * mw;
        a.tiles_x = (W + 31) / 32; a.tiles_y = (H + TH - 1) / TH;
        a.nct = L.cout / (32 * L.NW); a.nchunks = L.nchunks;
        double px = (double)nb * H * W;
        double flops = 2.0 * px * L.cout * (L.zt >= 0 ? t0.C : L.cin_real) * 9;
        double bytes = px * P * 2.0 * (t0.C + (L.upf ? c1 / 4.0 : c1) + (head ? 0 : L.cout)) + (L.do_pool ? px / 4 * P * 2.0 * L.cout : 0.0) + (double)L.cout * L.cin_real * 9 * 2.0 * P;
        if (L.zt >= 0) bytes += px * L.cout * 4.0 - (double)L.cout * (L.cin_real - t0.C) * 9 * 2.0 * P;
        if (L.c0f) {     // conv0_0.conv1 rides along: its flops, the raw input instead of x0_0a
          flops += 2.0 * px * 32 * 3 * 9;
          bytes += px * (in_format == UNETPP_IN_F32_NCHW ? 12.0 : 3.0) - px * P * 2.0 * t0.C;
        }
        if (head) {
          a.head_w = e->blob + e->head_w_off; a.head_b = e->blob + e->head_b_off; a.head_C = C;
          a.logits = lg; a.mask = mk; a.cable = cb; a.tape = tpe;
          a.probs = pr; a.rule = outp->rule;
          a.t_cable = outp->t_cable; a.t_tape = outp->t_tape; a.bg_margin = outp->bg_margin; a.ct_margin = outp->ct_margin;
          flops += 2.0 * px * 32 * C;
          bytes += px * ((lg ? 4.0 * C : 0) + (pr ? 4.0 * C : 0) + (mk ? 1 : 0) + (cb ? 1 : 0) + (tpe ? 1 : 0));
          head_done = true;
        }
        // exact mode, Cout = 32 or a multiple of 64, single source (or skip + fused upsample): the wave-specialised kernel
        const bool ws = layer_uses_ws(e, L);
        char lbl[160];
        // labels end in the kernel's full template argument list, as rocprofv3 prints it (bench.py matches on it)
        auto tf = [](bool v) { return v ? "true" : "false"; };
        if (ws) snprintf(lbl, sizeof lbl, "%s%s%s%s|conv3x3_ws_kernel<%d, %s, %s, %s, %s, %d, %d, %s, %s>", L.c0f ? "input+conv0_0.conv1+" : "", L.name.c_str(), L.upf ? "+up" : "", head ? "+final+argmax" : "", P, tf(L.do_pool), tf(head), tf(L.upf), tf(L.c0f), L.cout == 32 ? 1 : 2, L.cout == 32 ? 4 : 2, tf(e->x8), tf(L.in2 >= 0 && !L.upf));
        else snprintf(lbl, sizeof lbl, "%s%s%s|conv3x3_bias_relu_kernel<%d, %d, %d, %d, %d, %s, %s, %s, %s>", L.name.c_str(), L.upf ? "+up" : (L.zt >= 0 ? ".skip+z" : ""), head ? "+final+argmax" : "", P, L.KC, L.NW, mw, L.WAVES, tf(L.do_pool), tf(head), tf(L.upf), tf(L.zt >= 0));
        Lx.run(lbl, flops, bytes, [&] {
          return ws ? launch_ws(LaunchCtx{e->cfg.device, e->num_cus, e->ksplit_max, e->ksplit_min_chunks, e->ksplit_gate}, P, e->x8, a, L.do_pool, head, L.upf, L.c0f, s)
                    : launch_conv(LaunchCtx{e->cfg.device, e->num_cus}, P, L, mw, a, head, s);
        });
#ifdef UNETPP_WS_DBG
        if (stamp_this && ws) {
          static int printed = 0;
          std::vector<unsigned long long> hs(1024 * 32 * 5);
          (void)hipStreamSynchronize(s);
          (void)hipMemcpy(hs.data(), stamp_buf, hs.size() * 8, hipMemcpyDeviceToHost);
          if (printed++ == 3) {      // a warm launch
            double sum[32] = {0};
            int nwg = 0;
            for (int b = 0; b < 1024; ++b) { if (!hs[b * 32 + 1] && !hs[b * 32 + 16]) continue; ++nwg; for (int i = 0; i < 32; ++i) sum[i] += (double)hs[b * 32 + i]; }
            fprintf(stderr, "[stamps %s] %d workgroups, mean cycles per workgroup:\n  consumer: barrier %.0f  chunks %.0f  epilogue %.0f  init %.0f\n"
                            "  producer: skip issue %.0f wait %.0f barrier %.0f | up issue %.0f interp %.0f (reads %.0f arithmetic %.0f split+stores %.0f) wait %.0f barrier %.0f | setup %.0f\n", L.name.c_str(), nwg,
                    sum[0] / nwg, sum[1] / nwg, sum[2] / nwg, sum[3] / nwg, sum[16] / nwg, sum[17] / nwg, sum[18] / nwg, sum[19] / nwg, sum[20] / nwg, sum[24] / nwg, sum[25] / nwg, sum[26] / nwg, sum[21] / nwg, sum[22] / nwg, sum[23] / nwg);
            // wall-clock (100 MHz) begin / end of consumer wave 0 of every workgroup, relative to the earliest begin
            std::vector<double> bg, en;
            unsigned long long t0 = ~0ull;
            std::vector<double> cb, pe;
            for (int b = 0; b < 1024; ++b) if (hs[b * 32 + 6]) t0 = std::min(t0, hs[b * 32 + 6]);
            for (int b = 0; b < 1024; ++b) if (hs[b * 32 + 6]) {
              bg.push_back((hs[b * 32 + 6] - t0) * 0.01); en.push_back((hs[b * 32 + 5] - t0) * 0.01);
              cb.push_back((hs[b * 32 + 4] - t0) * 0.01); pe.push_back((hs[b * 32 + 16 + 12] - t0) * 0.01);
            }
            std::sort(bg.begin(), bg.end()); std::sort(en.begin(), en.end()); std::sort(cb.begin(), cb.end()); std::sort(pe.begin(), pe.end());
            if (!bg.empty()) fprintf(stderr, "  consumer loop begins p50 %.2f max %.2f | producer wave 0 ends p50 %.2f max %.2f\n", cb[cb.size() / 2], cb.back(), pe[pe.size() / 2], pe.back());
            if (a.dbg & 131072)
              for (int b : {0, 1, 100, 200}) {      // event logs of consumer wave 0 and producer wave 0 (phase:us after the earliest kernel entry)
                for (int role = 0; role < 2; ++role) {
                  fprintf(stderr, "  wg %3d %s:", b, role ? "producer" : "consumer");
                  const unsigned long long* ev = hs.data() + 1024 * 32 + ((size_t)b * 2 + role) * 64;
                  for (int i = 0; i < 60 && ev[i]; ++i) fprintf(stderr, " %d:%.2f", (int)(ev[i] >> 56), (double)((ev[i] & 0xffffffffffffffull) - t0) * 0.01);
                  fprintf(stderr, "\n");
                }
              }
            if (!bg.empty())
              fprintf(stderr, "  wall clock (us after the first workgroup's kernel entry): entry p50 %.2f p90 %.2f max %.2f | end min %.2f p50 %.2f p90 %.2f max %.2f\n",
                      bg[bg.size() / 2], bg[bg.size() * 9 / 10], bg.back(), en.front(), en[en.size() / 2], en[en.size() * 9 / 10], en.back());
          }
        }
#endif
      } else if (op.kind == OP_TAPMM || op.kind == OP_UPSUM) {
        const ConvLayer& L = e->convs[op.idx];
        const int H = h >> L.lvl, W = w >> L.lvl;
        const int cs = e->tensors[L.in].C, cup = L.cin_real - cs;
        if (op.kind == OP_TAPMM) {
          TapmmArgs t{};
          t.low = tp(L.low_t); t.wpk = L.tapw; t.y = (float*)tp(L.y_t);
          t.N = nb; t.hw = (H >> 1) * (W >> 1); t.K = cup; t.Nv = 9 * L.cout;
#ifdef UNETPP_WS_DBG
          { const char* d = getenv("UNETPP_TAPMM_DBG"); t.dbg = d ? atoi(d) : 0; }
#endif
          const int tiles = nb * ((t.hw + TapmmCfg::TM - 1) / TapmmCfg::TM) * (t.Nv / TapmmCfg::TN);
          const double M = (double)nb * t.hw;
          char lbl[96];
          snprintf(lbl, sizeof lbl, "%s.up-gemm|tapmm_ws_kernel<%s>", L.name.c_str(), e->x8 ? "true" : "false");
          Lx.run(lbl, 2.0 * M * cup * t.Nv, M * cup * 2.0 * P + M * t.Nv * 4.0 + (double)t.Nv * cup * 2.0 * P, [&] {
            auto k = e->x8 ? tapmm_ws_kernel<true> : tapmm_ws_kernel<false>;
            hipError_t st = allow_full_lds((const void*)k, e->cfg.device);
            if (st != hipSuccess) return st;
            hipLaunchKernelGGL(k, dim3((unsigned)std::min(tiles, e->num_cus)), dim3(TapmmCfg::NT), TapmmCfg::LDS_BYTES, s, t);
            return hipSuccess;
          });
        } else {
          UpsumArgs u{};
          u.y = (const float*)tp(L.y_t); u.z = (float*)tp(L.zt); u.N = nb; u.H = H; u.W = W; u.Cout = L.cout;
          unsigned blocks = (unsigned)(((W + UpsumCfg::TW - 1) / UpsumCfg::TW) * ((H + UpsumCfg::TH - 1) / UpsumCfg::TH) * (L.cout / 32) * nb);
          const double px = (double)nb * H * W;
          char lbl[96];
          const bool small = blocks <= (unsigned)e->num_cus;      // too few 16-row tiles for the chip: 4-row tiles
          if (small) blocks = (unsigned)(((W + 31) / 32) * ((H + 3) / 4) * (L.cout / 32) * nb);
          snprintf(lbl, sizeof lbl, "%s.up-sum|upsum_kernel<1, %d>", L.name.c_str(), small ? 4 : 16);
          Lx.run(lbl, px * L.cout * 9 * 8.0, px / 4 * 9 * L.cout * 4.0 + px * L.cout * 4.0, [&] {
            if (small) hipLaunchKernelGGL((upsum_kernel<1, 4>), dim3(blocks), dim3(256), UpsumCfgT<4>::LDS_BYTES, s, u);
            else hipLaunchKernelGGL((upsum_kernel<1, 16>), dim3(blocks), dim3(256), UpsumCfg::LDS_BYTES, s, u);
            return hipSuccess;
          });
        }
      } else if (op.kind == OP_UP) {
        const Tensor& low = e->tensors[op.idx];
        const Tensor& dst = e->tensors[op.out];
        const int H = h >> dst.lvl, W = w >> dst.lvl;
        double px = (double)nb * H * W;
        double bytes = px * P * 2.0 * low.C + px / 4 * P * 2.0 * low.C;
        const int nseg = (W + UP_SEG - 1) / UP_SEG;
        const int seg_w = std::min(W, UP_SEG);
        const size_t up_lds = (size_t)3 * (seg_w / 2 + 2) * P * 32;      // staged low-res rows of one segment
        const int up_threads = seg_w * 2 * P >= 1024 ? 512 : (seg_w * 2 * P >= 256 ? 256 : 128);   // one item = both rows of a piece
        char nm[64];
        snprintf(nm, sizeof nm, "up%d|upsample2x_kernel<%d>", dst.lvl, P);
        Lx.run(nm, px * low.C * 8, bytes, [&] {
          if (P == 2) hipLaunchKernelGGL(upsample2x_kernel<2>, dim3((unsigned)((H / 2) * nseg), (unsigned)(nb * (low.C / 16))), dim3(up_threads), up_lds, s, tp(op.idx), H, W, tp(op.out));
          else hipLaunchKernelGGL(upsample2x_kernel<1>, dim3((unsigned)((H / 2) * nseg), (unsigned)(nb * (low.C / 16))), dim3(up_threads), up_lds, s, tp(op.idx), H, W, tp(op.out));
          return hipSuccess;
        });
      } else if (op.kind == OP_CONVT) {
        ConvTLayer& T = e->convts[op.idx];
        ConvTArgs a{};
        const int H = h >> T.lvl, W = w >> T.lvl;
        a.in = tp(T.in); a.wpk = T.wpk; a.scale = T.scale; a.bias = e->blob + T.b_off; a.out = tp(T.out);
        a.N = nb; a.H = H; a.W = W; a.Cin = T.cin; a.Cout = T.cout;
        a.status = e->d_status;
        double px = (double)nb * H * W;
        double flops = 2.0 * px * T.cin * T.cout * 4;
        double bytes = px * P * 2.0 * (T.cin + 4.0 * T.cout) + 4.0 * T.cin * T.cout * 2.0 * P;
        dim3 grid((unsigned)(((H * W + 511) / 512) * (4 * T.cout / 64) * nb));
        char lbl[96];
        if (e->x8) snprintf(lbl, sizeof lbl, "%s|convt2x2_kernel<2, true>", T.name.c_str());
        else snprintf(lbl, sizeof lbl, "%s|convt2x2_kernel<%d>", T.name.c_str(), P);
        Lx.run(lbl, flops, bytes, [&] {
          if (e->x8) hipLaunchKernelGGL((convt2x2_kernel<2, true>), grid, dim3(256), 0, s, a);
          else if (P == 2) hipLaunchKernelGGL(convt2x2_kernel<2>, grid, dim3(256), 0, s, a);
          else hipLaunchKernelGGL(convt2x2_kernel<1>, grid, dim3(256), 0, s, a);
          return hipSuccess;
        });
      } else if (op.kind == OP_HEAD) {
        if (head_done) continue;     // ran in the last conv's epilogue
        const int cx = e->head_cx;
        size_t total = (size_t)nb * hw;
        double bytes = (double)total * (P * 2.0 * cx + (lg ? 4.0 * C : 0) + (pr ? 4.0 * C : 0) + (mk ? 1 : 0) + (cb ? 1 : 0) + (tpe ? 1 : 0));
        dim3 grid((unsigned)((hw + 255) / 256), (unsigned)nb);
        const size_t lds = (size_t)(C * cx + C) * sizeof(float);
        Lx.run(e->x8 ? "final+argmax|head_generic_kernel<3>" : P == 2 ? "final+argmax|head_generic_kernel<2>" : "final+argmax|head_generic_kernel<1>", 2.0 * total * cx * C, bytes, [&] {
          if (e->x8) hipLaunchKernelGGL(head_generic_kernel<3>, grid, dim3(256), lds, s, tp(op.out), cx, e->blob + e->head_w_off, e->blob + e->head_b_off, C, h, w, lg, pr, mk, cb, tpe, outp->rule, outp->t_cable, outp->t_tape, outp->bg_margin, outp->ct_margin);
          else if (P == 2) hipLaunchKernelGGL(head_generic_kernel<2>, grid, dim3(256), lds, s, tp(op.out), cx, e->blob + e->head_w_off, e->blob + e->head_b_off, C, h, w, lg, pr, mk, cb, tpe, outp->rule, outp->t_cable, outp->t_tape, outp->bg_margin, outp->ct_margin);
          else hipLaunchKernelGGL(head_generic_kernel<1>, grid, dim3(256), lds, s, tp(op.out), cx, e->blob + e->head_w_off, e->blob + e->head_b_off, C, h, w, lg, pr, mk, cb, tpe, outp->rule, outp->t_cable, outp->t_tape, outp->bg_margin, outp->ct_margin);
          return hipSuccess;
        });
      }
    }
    if (Lx.rc) return Lx.rc;
#ifdef UNETPP_WS_DBG
    if (!dbg_timeline.empty()) {
      // wall-clock (100 MHz, chip-wide) entry of the first workgroup and end of the last consumer-0 / producer-0 wave of every
      // wave-specialised launch, relative to the first launch's entry: what lies between the launches
      static int pass = 0;
      if (++pass == 6) {
        (void)hipStreamSynchronize(s);
        std::vector<unsigned long long> hs((size_t)(1 + dbg_timeline.size()) * 1024 * 32);
        (void)hipMemcpy(hs.data(), unetpp_dbg_stamp_buf, hs.size() * 8, hipMemcpyDeviceToHost);
        unsigned long long T0 = ~0ull;
        double prev_end = 0;
        for (size_t li = 0; li < dbg_timeline.size(); ++li) {
          const unsigned long long* r = hs.data() + (1 + li) * 1024 * 32;
          unsigned long long first = ~0ull, last = 0; int nwg = 0;
          for (int b = 0; b < 1024; ++b) if (r[b * 32 + 6]) { ++nwg; first = std::min(first, r[b * 32 + 6]); last = std::max(last, std::max(r[b * 32 + 5], r[b * 32 + 16 + 12])); }
          if (!nwg) continue;
          if (T0 == ~0ull) T0 = first;
          const double b0 = (first - T0) * 0.01, e0 = (last - T0) * 0.01;
          fprintf(stderr, "[timeline] %-16s %4d wg  entry %8.2f  end %8.2f  inside %6.2f  since previous end %6.2f\n", dbg_timeline[li].c_str(), nwg, b0, e0, e0 - b0, b0 - prev_end);
          prev_end = e0;
        }
      }
    }
#endif
  }
  HIP_TRY(e, join.run());
  e->kcnt_dirty = Lx.rc != UNETPP_OK;
  return Lx.rc;
}

int unetpp_status(unetpp_engine* e, uint32_t* flags, int clear) {
  if (!e) return UNETPP_E_INVALID;
  if (!flags) return fail(e, UNETPP_E_INVALID, "flags is NULL");
  ENTER_DEVICE(e);
  HIP_TRY(e, hipDeviceSynchronize());      // every forward queued so far has set its bits
  unsigned v = 0;
  HIP_TRY(e, hipMemcpy(&v, e->d_status, sizeof v, hipMemcpyDeviceToHost));
  if (clear && v) HIP_TRY(e, hipMemset(e->d_status, 0, sizeof v));
  *flags = v;
  return UNETPP_OK;
}

int unetpp_mask_stats(unetpp_engine* e, const uint8_t* dev_mask, int batch, int h, int w, uint32_t* dev_counts,
                      int32_t* dev_row_min, int32_t* dev_row_max, void* stream) {
  if (!e) return UNETPP_E_INVALID;
  if (!dev_mask || !dev_counts || !dev_row_min || !dev_row_max) return fail(e, UNETPP_E_INVALID, "null argument");
  if (batch < 1 || h < 1 || w < 1) return fail(e, UNETPP_E_INVALID, "bad shape %dx%dx%d", batch, h, w);
  ENTER_DEVICE(e);
  hipStream_t s = (hipStream_t)stream;
  const int C = e->cfg.num_classes;
  HIP_TRY(e, hipMemsetAsync(dev_counts, 0, (size_t)batch * C * sizeof(uint32_t), s));
  hipLaunchKernelGGL(mask_stats_kernel, dim3((unsigned)h, (unsigned)batch), dim3(256), 0, s, dev_mask, C, h, w,
                     (unsigned*)dev_counts, (int*)dev_row_min, (int*)dev_row_max);
  HIP_TRY(e, hipGetLastError());
  return UNETPP_OK;
}

// ---- frame glue: cv2.resize either side of the model ----------------------------------
}  // extern "C"
namespace {
// resizeGeneric_'s INTER_LINEAR index/coefficient tables (OpenCV imgproc/src/resize.cpp): double index math, float
// fraction, saturate_cast<short>(c * 2048) with round-half-even.  Entry = {s0, s1, a0, a1}.
void linear_table(int n_src, int n_dst, std::vector<int>& t) {
#pragma clang fp contract(off)
  t.resize((size_t)n_dst * 4);
  const double inv_scale = (double)n_dst / (double)n_src;
  const double scale = 1.0 / inv_scale;
  for (int d = 0; d < n_dst; ++d) {
    float fx = (float)((d + 0.5) * scale - 0.5);
    int s0 = (int)floorf(fx);
    fx -= (float)s0;
    if (s0 < 0) { fx = 0.f; s0 = 0; }
    if (s0 >= n_src - 1) { fx = 0.f; s0 = n_src - 1; }
    const float c0 = (1.f - fx) * 2048.f, c1 = fx * 2048.f;
    long a0 = lrintf(c0), a1 = lrintf(c1);
    a0 = std::min(32767L, std::max(-32768L, a0));
    a1 = std::min(32767L, std::max(-32768L, a1));
    t[4 * d + 0] = s0; t[4 * d + 1] = std::min(s0 + 1, n_src - 1); t[4 * d + 2] = (int)a0; t[4 * d + 3] = (int)a1;
  }
}
// resizeNN's index table: min(floor(d * (1 / (n_dst / n_src))), n_src - 1) in double.
void nearest_table(int n_src, int n_dst, std::vector<int>& t) {
#pragma clang fp contract(off)
  t.resize((size_t)n_dst);
  const double inv = (double)n_dst / (double)n_src;
  const double ifx = 1.0 / inv;
  for (int d = 0; d < n_dst; ++d) t[d] = std::min((int)floor(d * ifx), n_src - 1);
}
// Device copy of a table, built on first use (that first call synchronises: a blocking hipMemcpy).
int resize_table(unetpp_engine* e, int kind, int n_src, int n_dst, void** out) {
  auto key = std::make_tuple(kind, n_src, n_dst);
  auto it = e->resize_tabs.find(key);
  if (it == e->resize_tabs.end()) {
    std::vector<int> t;
    if (kind == 0) linear_table(n_src, n_dst, t); else nearest_table(n_src, n_dst, t);
    void* d = nullptr;
    HIP_TRY(e, hipMalloc(&d, t.size() * sizeof(int)));
    hipError_t r = hipMemcpy(d, t.data(), t.size() * sizeof(int), hipMemcpyHostToDevice);
    if (r != hipSuccess) { (void)hipFree(d); return fail(e, UNETPP_E_HIP, "hipMemcpy(resize table): %s", hipGetErrorString(r)); }
    it = e->resize_tabs.emplace(key, d).first;
  }
  *out = it->second;
  return UNETPP_OK;
}
}  // namespace
extern "C" {

int unetpp_resize_linear_u8(unetpp_engine* e, const uint8_t* dev_src, int batch, int src_h, int src_w, int channels,
                            uint8_t* dev_dst, int dst_h, int dst_w, void* stream) {
  if (!e) return UNETPP_E_INVALID;
  if (!dev_src || !dev_dst) return fail(e, UNETPP_E_INVALID, "null argument");
  if (batch < 1 || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1 || channels < 1 || channels > 4)
    return fail(e, UNETPP_E_INVALID, "bad resize shape %dx%dx%dx%d -> %dx%d", batch, src_h, src_w, channels, dst_h, dst_w);
  if (batch > 65535 || dst_h > 65535 || (size_t)src_h * src_w * channels > 0x7fffffffULL)
    return fail(e, UNETPP_E_INVALID, "resize shape too large");
  ENTER_DEVICE(e);
  void *xt = nullptr, *yt = nullptr;
  int rc = resize_table(e, 0, src_w, dst_w, &xt); if (rc) return rc;
  rc = resize_table(e, 0, src_h, dst_h, &yt); if (rc) return rc;
  const unsigned gx = (unsigned)((dst_w * channels + 1023) / 1024);
  hipLaunchKernelGGL(resize_linear_u8_kernel, dim3(gx, (unsigned)dst_h, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                     dev_src, src_h, src_w, channels, dev_dst, dst_h, dst_w, (const int4*)xt, (const int4*)yt);
  HIP_TRY(e, hipGetLastError());
  return UNETPP_OK;
}

int unetpp_resize_nearest_roi_u8(unetpp_engine* e, const uint8_t* dev_src, int batch, int src_h, int src_w, int match_class,
                                 uint8_t* dev_dst, int dst_h, int dst_w, int x1, int y1, int x2, int y2, void* stream) {
  if (!e) return UNETPP_E_INVALID;
  if (!dev_src || !dev_dst) return fail(e, UNETPP_E_INVALID, "null argument");
  if (batch < 1 || src_h < 1 || src_w < 1 || dst_h < 1 || dst_w < 1)
    return fail(e, UNETPP_E_INVALID, "bad resize shape %dx%dx%d -> %dx%d", batch, src_h, src_w, dst_h, dst_w);
  if (batch > 65535 || dst_h > 65535) return fail(e, UNETPP_E_INVALID, "resize shape too large");
  if (x1 < 0 || y1 < 0 || x2 < 0 || y2 < 0) return fail(e, UNETPP_E_INVALID, "negative ROI bound (%d, %d, %d, %d)", x1, y1, x2, y2);
  if (match_class > 255) return fail(e, UNETPP_E_INVALID, "match_class %d out of range", match_class);
  ENTER_DEVICE(e);
  void *xo = nullptr, *yo = nullptr;
  int rc = resize_table(e, 1, src_w, dst_w, &xo); if (rc) return rc;
  rc = resize_table(e, 1, src_h, dst_h, &yo); if (rc) return rc;
  const unsigned gx = (unsigned)((dst_w + 1023) / 1024);
  hipLaunchKernelGGL(resize_nearest_roi_u8_kernel, dim3(gx, (unsigned)dst_h, (unsigned)batch), dim3(256), 0, (hipStream_t)stream,
                     dev_src, src_h, src_w, dev_dst, dst_h, dst_w, (const int*)xo, (const int*)yo, match_class, x1, y1, x2, y2);
  HIP_TRY(e, hipGetLastError());
  return UNETPP_OK;
}

// ---- profiling -------------------------------------------------------------------------------------
int unetpp_profile_enable(unetpp_engine* e, int on) {
  if (!e) return UNETPP_E_INVALID;
  e->prof_on = on != 0;
  e->prof_used = 0;
  e->ev_used = 0;
  e->prof_prev_ev = -1;
  return UNETPP_OK;
}
int unetpp_profile_count(const unetpp_engine* e) { return e ? e->prof_used : 0; }
int unetpp_profile_read(unetpp_engine* e, float* ms_out, int n) {
  if (!e || !ms_out) return UNETPP_E_INVALID;
  ENTER_DEVICE(e);
  int m = std::min(n, e->prof_used);
  for (int i = 0; i < m; ++i) {
    HIP_TRY(e, hipEventSynchronize(e->evpool[e->prof[i].ev1]));
    HIP_TRY(e, hipEventElapsedTime(&ms_out[i], e->evpool[e->prof[i].ev0], e->evpool[e->prof[i].ev1]));
  }
  return m;
}
const char* unetpp_profile_name(const unetpp_engine* e, int i) {
  if (!e || i < 0 || i >= e->prof_used) return "";
  return e->prof[i].name.c_str();
}
int unetpp_profile_work(const unetpp_engine* e, int i, double* flops, double* bytes) {
  if (!e || i < 0 || i >= e->prof_used) return UNETPP_E_INVALID;
  if (flops) *flops = e->prof[i].flops;
  if (bytes) *bytes = e->prof[i].bytes;
  return UNETPP_OK;
}

// ---- debug -------------------------------------------------------------------------------------
int unetpp_debug_keep_intermediates(unetpp_engine* e, int on) {
  if (!e) return UNETPP_E_INVALID;
  e->keep_all = on != 0;
  return UNETPP_OK;
}

long long unetpp_debug_read(unetpp_engine* e, const char* name, float* host_out, size_t max_floats) {
  if (!e || !name || !host_out) return UNETPP_E_INVALID;
  if (e->last_b == 0) return fail(e, UNETPP_E_STATE, "debug_read before forward");
  const Tensor* t = nullptr;
  int tid = -1;
  // "name#hi", "name#lo", "name#x8": one stored plane instead of the reconstructed value (see unpack_nchw_kernel)
  std::string base(name);
  int plane = 0;
  if (size_t hash = base.find('#'); hash != std::string::npos) {
    const std::string sel = base.substr(hash + 1);
    base.resize(hash);
    plane = sel == "hi" ? 1 : sel == "lo" ? 2 : sel == "x8" ? 3 : -1;
    if (plane < 0 || (plane == 3 && !e->x8) || (plane == 2 && e->P != 2))
      return fail(e, UNETPP_E_INVALID, "plane '%s' does not exist in this engine's activation format", sel.c_str());
  }
  for (size_t i = 0; i < e->tensors.size(); ++i)
    if (e->tensors[i].name == base) { t = &e->tensors[i]; tid = (int)i; }
  if (!t) return fail(e, UNETPP_E_INVALID, "unknown tensor '%s'", base.c_str());
  if (t->virt) return fail(e, UNETPP_E_STATE, "'%s' is never materialised on this engine (fused into its consumer)", name);
  if (t->C == 0) return fail(e, UNETPP_E_STATE, "'%s' is not an activation tensor (raw fp32 / bookkeeping buffer)", name);
  if (e->cfg.arch == UNETPP_ARCH_NESTED && tid == e->t_head_in && !e->keep_all)
    return fail(e, UNETPP_E_STATE, "x0_4 is not materialised (head fused): call unetpp_debug_keep_intermediates(e, 1) before forward");
  ENTER_DEVICE(e);
  int nb = e->last_b % e->mb == 0 ? std::min(e->mb, e->last_b) : e->last_b % e->mb;
  const int H = e->last_h >> t->lvl, W = e->last_w >> t->lvl;
  size_t total = (size_t)nb * t->C * H * W;
  if (total > max_floats) return fail(e, UNETPP_E_INVALID, "buffer too small: need %zu floats", total);
  float* tmp = nullptr;
  HIP_TRY(e, hipDeviceSynchronize());
  HIP_TRY(e, hipMalloc((void**)&tmp, total * sizeof(float)));
  const half_t* src = (const half_t*)(e->arena + e->last_slot_off + t->off);
  if (e->x8) hipLaunchKernelGGL(unpack_nchw_kernel<3>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, src, nb, t->C, H, W, tmp, plane);
  else if (e->P == 2) hipLaunchKernelGGL(unpack_nchw_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, src, nb, t->C, H, W, tmp, plane);
  else hipLaunchKernelGGL(unpack_nchw_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, src, nb, t->C, H, W, tmp, plane);
  hipError_t st = hipMemcpy(host_out, tmp, total * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (st != hipSuccess) return fail(e, UNETPP_E_HIP, "debug copy: %s", hipGetErrorString(st));
  return (long long)total;
}

}  // extern "C"
