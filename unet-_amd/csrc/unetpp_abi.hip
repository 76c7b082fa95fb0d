// unetpp_abi.hip — engine + C ABI (include/unetpp.h) of the MI355X-native UNet++ inference path.
// Graph executed (reference src/models/unetpp.py:104-119, eval mode):
//   x0_0 = CB(3,32)(x)            x1_0 = CB(32,64)(pool x0_0)     x2_0 = CB(64,128)(pool x1_0)
//   x3_0 = CB(128,256)(pool x2_0) x4_0 = CB(256,512)(pool x3_0)
//   x3_1 = CB(768,256)(cat[x3_0, up x4_0])   x2_2 = CB(384,128)(cat[x2_0, up x3_1])
//   x1_3 = CB(192,64)(cat[x1_0, up x2_2])    x0_4 = CB(96,32)(cat[x0_0, up x1_3])
//   out  = Conv1x1(32,C)(x0_4)  -> argmax / class masks (infer_two_stage_burr.py:299-304)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/unetpp.h"
#include "aux_kernels.h"
#include "conv3x3_mfma.h"

using namespace unetpp;

namespace {

thread_local std::string g_create_error;

constexpr uint32_t BLOB_MAGIC = 0x50504e55u;  // 'UNPP'
constexpr int BLOB_VERSION = 1;
const int NB[5] = {32, 64, 128, 256, 512};    // nb_filter, reference unetpp.py:49

struct Tensor {
  half_t* p = nullptr;
  int C = 0;
  int lvl = 0;
};

struct ConvLayer {
  std::string name;
  int cin_real = 0;   // channels the canonical weight has
  int cout = 0;
  int lvl = 0;
  Tensor in, in2, out, pool;   // in2: second source of the virtual concat (C = 0 when unused)
  bool do_pool = false;
  size_t w_off = 0, b_off = 0;  // float offsets inside the canonical blob payload
  int KC = 16, NW = 1, MW = 2, WAVES = 8;
  int nchunks = 1;
  half_t* wpk = nullptr;
  float* scale = nullptr;
  float* mult = nullptr;
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
  int mb = 1;
  std::string err;
  char* arena = nullptr;
  size_t arena_bytes = 0;
  float* blob = nullptr;  // canonical fp32 blob payload on device (weights + biases)
  size_t blob_floats = 0;
  bool weights_loaded = false;
  std::vector<ConvLayer> convs;  // 18
  Tensor in8, up[4];             // up[l]: bilinear x2 of the level l+1 node, at level l
  Tensor x[5], xa[5], pooled[4], d[4], da[4];  // encoder x{l}_0, its conv1 temp, pooled; decoder nodes
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
  bool keep_all = false;   // debug: materialise x0_4 and run the head as its own kernel
  // concurrent micro-batches: `nstreams` copies of the activation area, one internal stream each
  int nstreams = 1;
  size_t act_bytes = 0;          // size of one activation area (slot)
  size_t last_slot_off = 0;
  hipStream_t streams[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_start = nullptr, ev_done[4] = {nullptr, nullptr, nullptr, nullptr};
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

size_t blob_payload_floats(int C, int cin) {
  size_t n = 0;
  int ci[9] = {cin, NB[0], NB[1], NB[2], NB[3], NB[3] + NB[4], NB[2] + NB[3], NB[1] + NB[2], NB[0] + NB[1]};
  int co[9] = {NB[0], NB[1], NB[2], NB[3], NB[4], NB[3], NB[2], NB[1], NB[0]};
  for (int b = 0; b < 9; ++b) {
    n += (size_t)co[b] * ci[b] * 9 + co[b];
    n += (size_t)co[b] * co[b] * 9 + co[b];
  }
  n += (size_t)C * NB[0] + C;
  return n;
}

// ---- conv dispatch ---------------------------------------------------------------------------
int g_num_cus = 256;

template <int P, int KC, int NW, int MW, int WAVES, bool POOL, bool HEAD>
hipError_t launch_conv_k(const ConvArgs& a, hipStream_t s) {
  using C = ConvCfg<P, KC, NW, MW, WAVES>;
  const int lds = C::LDS_BYTES + a.Cout * 8 + (HEAD ? ((a.head_C * 33 * 4 + 15) / 16) * 16 : 0);
  // persistent workgroups: as many as are resident at once, each walks tiles blockIdx, +grid, ...
  const int total = a.N * a.tiles_x * a.tiles_y * a.nct;
  const int per_cu = std::max(1, std::min(2, (160 * 1024) / lds));
  dim3 grid((unsigned)std::min(total, g_num_cus * per_cu));
  auto k = conv3x3_bias_relu_kernel<P, KC, NW, MW, WAVES, POOL, HEAD>;
  static int attr_lds = 0;
  if (attr_lds < lds) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds); attr_lds = lds; }
  hipLaunchKernelGGL(k, grid, dim3(C::NT), lds, s, a);
  return hipGetLastError();
}

template <int P, int KC, int NW, int MW, int WAVES>
hipError_t launch_conv_cfg(const ConvArgs& a, bool pool, bool head, hipStream_t s) {
  if constexpr (NW == 1) {
    if (head) return launch_conv_k<P, KC, NW, MW, WAVES, false, true>(a, s);
  }
  if (pool) return launch_conv_k<P, KC, NW, MW, WAVES, true, false>(a, s);
  return launch_conv_k<P, KC, NW, MW, WAVES, false, false>(a, s);
}

hipError_t launch_conv(int P, const ConvLayer& L, const ConvArgs& a, bool head, hipStream_t s) {
#define CASE(p, kc, nw, mw, wv) \
  if (P == p && L.KC == kc && L.NW == nw && L.MW == mw && L.WAVES == wv) return launch_conv_cfg<p, kc, nw, mw, wv>(a, L.do_pool, head, s);
  CASE(1, 16, 1, 2, 8)
  CASE(1, 32, 1, 2, 4)
  CASE(1, 32, 2, 2, 8)
  CASE(1, 16, 4, 2, 8)
  CASE(2, 16, 1, 2, 8)
  CASE(2, 16, 2, 2, 8)
#undef CASE
  return hipErrorInvalidValue;
}

void choose_cfg(int P, ConvLayer& L) {
  L.MW = 2; L.WAVES = 8;
  const int cin = L.in.C + L.in2.C;
  if (P == 1) {
    if (L.cout == 32) { L.NW = 1; L.KC = (cin <= 16) ? 16 : 32; if (L.KC == 32) L.WAVES = 4; }
    else if (L.cout == 64) { L.NW = 2; L.KC = 32; }
    else { L.NW = 4; L.KC = 16; }
  } else {
    L.KC = 16;
    L.NW = (L.cout == 32) ? 1 : 2;
  }
  L.nchunks = (L.in.C + L.KC - 1) / L.KC + L.in2.C / L.KC;
}

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace

namespace {
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
        hipEvent_t ev;
        (void)hipEventCreate(&ev);
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
        (void)hipEventRecord(e->evpool[r->ev0], s);
      }
    }
    hipError_t st = f();
    if (st == hipSuccess) st = hipGetLastError();
    if (r) {
      r->ev1 = new_event();
      (void)hipEventRecord(e->evpool[r->ev1], s);
      e->prof_prev_ev = r->ev1; e->prof_prev_stream = s;
    }
    if (st != hipSuccess) rc = fail(e, UNETPP_E_HIP, "launch %s: %s", name.c_str(), hipGetErrorString(st));
  }
};
}  // namespace

extern "C" {

const char* unetpp_version(void) { return "unetpp-hip 0.1.0 (gfx950)"; }

const char* unetpp_last_error(const unetpp_engine* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

size_t unetpp_weights_blob_bytes(int num_classes, int in_channels) {
  return 32 + 4 * blob_payload_floats(num_classes, in_channels);
}

int unetpp_create(const unetpp_config* cfg, unetpp_engine** out) {
  if (!cfg || !out) return fail(nullptr, UNETPP_E_INVALID, "null argument");
  *out = nullptr;
  if (cfg->in_channels != 3) return fail(nullptr, UNETPP_E_UNSUPPORTED, "input_channels=%d unsupported (only 3)", cfg->in_channels);
  if (cfg->num_classes < 1 || cfg->num_classes > HEAD_MAX_CLASSES)
    return fail(nullptr, UNETPP_E_INVALID, "num_classes=%d out of range [1,%d]", cfg->num_classes, HEAD_MAX_CLASSES);
  if (cfg->max_batch < 1 || cfg->max_h < 16 || cfg->max_w < 16 || cfg->max_h % 16 || cfg->max_w % 16)
    return fail(nullptr, UNETPP_E_INVALID, "max shape (%d,%d,%d): batch>=1 and H,W positive multiples of 16 required",
                cfg->max_batch, cfg->max_h, cfg->max_w);
  if (cfg->precision != UNETPP_PREC_EXACT && cfg->precision != UNETPP_PREC_FAST)
    return fail(nullptr, UNETPP_E_INVALID, "precision=%d unknown", cfg->precision);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, UNETPP_E_HIP, "no HIP device available: this engine has no CPU fallback");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, UNETPP_E_INVALID, "device %d not in [0,%d)", cfg->device, ndev);
  HIP_TRY(nullptr, hipSetDevice(cfg->device));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) g_num_cus = prop.multiProcessorCount;
  }

  unetpp_engine* e = new unetpp_engine();
  e->cfg = *cfg;
  e->P = cfg->precision == UNETPP_PREC_EXACT ? 2 : 1;
  e->mb = (cfg->micro_batch > 0 && cfg->micro_batch < cfg->max_batch) ? cfg->micro_batch : cfg->max_batch;
  e->nstreams = std::max(1, std::min(4, cfg->streams));
  if (e->mb >= cfg->max_batch) e->nstreams = 1;      // a single pass has nothing to overlap with
  const int P = e->P;

  // ---- tensor plan (sizes for micro-batch x max_h x max_w)
  std::vector<std::pair<Tensor*, size_t>> allocs;
  size_t total = 0;
  auto plan = [&](Tensor& t, int C, int lvl) {
    t.C = C; t.lvl = lvl;
    size_t px = (size_t)e->mb * (cfg->max_h >> lvl) * (cfg->max_w >> lvl);
    size_t bytes = align_up(px * P * C * sizeof(half_t), 256);
    allocs.push_back({&t, total});
    total += bytes;
  };
  plan(e->in8, 8, 0);
  for (int l = 0; l < 5; ++l) { plan(e->xa[l], NB[l], l); plan(e->x[l], NB[l], l); }
  for (int l = 0; l < 4; ++l) plan(e->pooled[l], NB[l], l + 1);
  for (int l = 0; l < 4; ++l) { plan(e->up[l], NB[l + 1], l); plan(e->da[l], NB[l], l); plan(e->d[l], NB[l], l); }

  e->act_bytes = align_up(total, 4096);
  total = e->act_bytes * e->nstreams;

  // ---- conv layers in forward order; canonical blob offsets
  size_t off = 0;
  Tensor none;
  auto add = [&](const std::string& name, int cin_real, const Tensor& in, const Tensor& in2, const Tensor& outT, int lvl, bool pool) {
    ConvLayer L;
    L.name = name; L.cin_real = cin_real; L.in = in; L.in2 = in2; L.out = outT; L.cout = outT.C; L.lvl = lvl;
    L.do_pool = pool;
    L.w_off = off; off += (size_t)L.cout * cin_real * 9;
    L.b_off = off; off += L.cout;
    choose_cfg(P, L);
    e->convs.push_back(L);
  };
  for (int l = 0; l < 5; ++l) {
    char nm[32];
    snprintf(nm, sizeof nm, "conv%d_0", l);
    add(std::string(nm) + ".conv1", l == 0 ? 3 : NB[l - 1], l == 0 ? e->in8 : e->pooled[l - 1], none, e->xa[l], l, false);
    add(std::string(nm) + ".conv2", NB[l], e->xa[l], none, e->x[l], l, l < 4);
  }
  for (int l = 3; l >= 0; --l) {
    char nm[32];
    snprintf(nm, sizeof nm, "conv%d_%d", l, 4 - l);
    add(std::string(nm) + ".conv1", NB[l] + NB[l + 1], l == 3 ? e->x[3] : e->x[l], e->up[l], e->da[l], l, false);
    add(std::string(nm) + ".conv2", NB[l], e->da[l], none, e->d[l], l, false);
  }
  e->head_w_off = off; off += (size_t)cfg->num_classes * NB[0];
  e->head_b_off = off; off += cfg->num_classes;
  e->blob_floats = off;
  if (off != blob_payload_floats(cfg->num_classes, 3)) { delete e; return fail(nullptr, UNETPP_E_STATE, "internal: blob size mismatch"); }

  // ---- packed weights + scales + blob in the same arena
  size_t blob_off = total; total += align_up(off * sizeof(float), 256);
  std::vector<size_t> wpk_off(e->convs.size()), sc_off(e->convs.size()), mu_off(e->convs.size());
  for (size_t i = 0; i < e->convs.size(); ++i) {
    ConvLayer& L = e->convs[i];
    size_t wb = (size_t)(L.cout / (32 * L.NW)) * L.nchunks * P * 9 * L.KC * (32 * L.NW) * sizeof(half_t);
    wpk_off[i] = total; total += align_up(wb, 256);
    sc_off[i] = total; total += align_up(L.cout * sizeof(float), 256);
    mu_off[i] = total; total += align_up(L.cout * sizeof(float), 256);
  }
  hipError_t st = hipMalloc((void**)&e->arena, total);
  if (st != hipSuccess) {
    std::string m = hipGetErrorString(st);
    delete e;
    return fail(nullptr, UNETPP_E_HIP, "hipMalloc(%zu bytes): %s", total, m.c_str());
  }
  e->arena_bytes = total;
  for (auto& a : allocs) a.first->p = (half_t*)(e->arena + a.second);
  e->blob = (float*)(e->arena + blob_off);
  for (size_t i = 0; i < e->convs.size(); ++i) {
    e->convs[i].wpk = (half_t*)(e->arena + wpk_off[i]);
    e->convs[i].scale = (float*)(e->arena + sc_off[i]);
    e->convs[i].mult = (float*)(e->arena + mu_off[i]);
  }
  // tensor pointers were copied into layers before allocation: rebind
  {
    size_t i = 0;
    for (int l = 0; l < 5; ++l) {
      e->convs[i].in = (l == 0) ? e->in8 : e->pooled[l - 1]; e->convs[i].out = e->xa[l]; ++i;
      e->convs[i].in = e->xa[l]; e->convs[i].out = e->x[l]; if (l < 4) e->convs[i].pool = e->pooled[l]; ++i;
    }
    for (int l = 3; l >= 0; --l) {
      e->convs[i].in = e->x[l]; e->convs[i].in2 = e->up[l]; e->convs[i].out = e->da[l]; ++i;
      e->convs[i].in = e->da[l]; e->convs[i].out = e->d[l]; ++i;
    }
  }
  if (e->nstreams > 1) {
    for (int i = 0; i < e->nstreams; ++i) {
      HIP_TRY(nullptr, hipStreamCreateWithFlags(&e->streams[i], hipStreamNonBlocking));
      HIP_TRY(nullptr, hipEventCreateWithFlags(&e->ev_done[i], hipEventDisableTiming));
    }
    HIP_TRY(nullptr, hipEventCreateWithFlags(&e->ev_start, hipEventDisableTiming));
  }
  *out = e;
  return UNETPP_OK;
}

void unetpp_destroy(unetpp_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  for (auto ev : e->evpool) (void)hipEventDestroy(ev);
  for (int i = 0; i < 4; ++i) { if (e->streams[i]) (void)hipStreamDestroy(e->streams[i]); if (e->ev_done[i]) (void)hipEventDestroy(e->ev_done[i]); }
  if (e->ev_start) (void)hipEventDestroy(e->ev_start);
  if (e->arena) (void)hipFree(e->arena);
  delete e;
}

size_t unetpp_workspace_bytes(const unetpp_engine* e) { return e ? e->arena_bytes : 0; }

static int repack(unetpp_engine* e, hipStream_t s) {
  const int P = e->P;
  for (auto& L : e->convs) {
    const float* w = e->blob + L.w_off;
    hipLaunchKernelGGL(weight_scale_kernel, dim3(L.cout), dim3(256), 0, s, w, L.cin_real * 9, L.mult, L.scale);
    const int BN = 32 * L.NW;
    long long units = (long long)(L.cout / BN) * L.nchunks * P * 9 * (L.KC / 8) * BN;
    hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, s, w, L.mult, L.cin_real,
                       L.cout, P, L.KC, BN, L.nchunks, L.wpk, units);
  }
  HIP_TRY(e, hipGetLastError());
  e->weights_loaded = true;
  return UNETPP_OK;
}

static int check_header(unetpp_engine* e, const uint32_t* h, size_t bytes) {
  if (bytes != unetpp_weights_blob_bytes(e->cfg.num_classes, e->cfg.in_channels))
    return fail(e, UNETPP_E_INVALID, "weight blob is %zu bytes, expected %zu for num_classes=%d", bytes,
                unetpp_weights_blob_bytes(e->cfg.num_classes, e->cfg.in_channels), e->cfg.num_classes);
  if (h[0] != BLOB_MAGIC || (int)h[1] != BLOB_VERSION) return fail(e, UNETPP_E_INVALID, "bad weight blob magic/version");
  if ((int)h[2] != e->cfg.num_classes || (int)h[3] != e->cfg.in_channels || (int)h[4] != 19)
    return fail(e, UNETPP_E_INVALID, "weight blob is for num_classes=%u in_channels=%u convs=%u", h[2], h[3], h[4]);
  return UNETPP_OK;
}

int unetpp_load_weights(unetpp_engine* e, const void* host_blob, size_t bytes) {
  if (!e || !host_blob) return fail(e, UNETPP_E_INVALID, "null argument");
  HIP_TRY(e, hipSetDevice(e->cfg.device));
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
  HIP_TRY(e, hipSetDevice(e->cfg.device));
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
  const bool want_probs = dev_probs != nullptr || outp->rule != UNETPP_RULE_ARGMAX;
  if (outp->rule != UNETPP_RULE_ARGMAX && e->cfg.num_classes < 3)
    return fail(e, UNETPP_E_INVALID, "class rules need num_classes >= 3 (bg, cable, tape)");
  if (want_probs && (e->keep_all || e->cfg.num_classes > HEAD_FUSED_MAX_CLASSES))
    return fail(e, UNETPP_E_UNSUPPORTED, "probabilities / class rules run only in the fused head (num_classes <= %d, debug keep off)", HEAD_FUSED_MAX_CLASSES);
  if (!dev_input) return fail(e, UNETPP_E_INVALID, "input is NULL");
  if (!e->weights_loaded) return fail(e, UNETPP_E_STATE, "forward before load_weights");
  if (in_format != UNETPP_IN_F32_NCHW && in_format != UNETPP_IN_U8_NHWC_BGR) return fail(e, UNETPP_E_INVALID, "unknown input format %d", in_format);
  if (batch < 1 || batch > e->cfg.max_batch) return fail(e, UNETPP_E_INVALID, "batch %d not in [1,%d]", batch, e->cfg.max_batch);
  if (h < 16 || w < 16 || h % 16 || w % 16)
    return fail(e, UNETPP_E_INVALID, "Sizes of tensors must match: H=%d W=%d must be positive multiples of 16", h, w);
  if (h > e->cfg.max_h || w > e->cfg.max_w) return fail(e, UNETPP_E_INVALID, "shape %dx%d exceeds engine maximum %dx%d", h, w, e->cfg.max_h, e->cfg.max_w);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  hipStream_t user_stream = (hipStream_t)stream;
  hipStream_t s = user_stream;
  const int P = e->P, C = e->cfg.num_classes;
  const bool multi = e->nstreams > 1 && batch > e->mb;
  if (multi) {   // fork: the internal streams start after everything already queued on the caller's stream
    HIP_TRY(e, hipEventRecord(e->ev_start, user_stream));
    for (int i = 0; i < e->nstreams; ++i) HIP_TRY(e, hipStreamWaitEvent(e->streams[i], e->ev_start, 0));
  }
  e->last_b = batch; e->last_h = h; e->last_w = w;
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
    auto sp = [&](const Tensor& t) { return (half_t*)((char*)t.p + slot_off); };
    // 1. input conversion
    {
      const char* src = (const char*)dev_input + (in_format == UNETPP_IN_F32_NCHW ? (size_t)b0 * 3 * hw * 4 : (size_t)b0 * hw * 3);
      size_t total = (size_t)nb * hw;
      double bytes = (double)total * (in_format == UNETPP_IN_F32_NCHW ? 12 : 3) + (double)total * P * 16;
      Lx.run(P == 2 ? "convert_input|convert_input_kernel<2>" : "convert_input|convert_input_kernel<1>", 0, bytes, [&] {
        if (P == 2) hipLaunchKernelGGL(convert_input_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const void*)src, in_format, nb, h, w, sp(e->in8));
        else hipLaunchKernelGGL(convert_input_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const void*)src, in_format, nb, h, w, sp(e->in8));
        return hipSuccess;
      });
    }
    float* lg = dev_logits ? dev_logits + (size_t)b0 * C * hw : nullptr;
    uint8_t* mk = dev_mask ? dev_mask + (size_t)b0 * hw : nullptr;
    uint8_t* cb = dev_cable ? dev_cable + (size_t)b0 * hw : nullptr;
    uint8_t* tp = dev_tape ? dev_tape + (size_t)b0 * hw : nullptr;
    float* pr = dev_probs ? dev_probs + (size_t)b0 * C * hw : nullptr;
    const bool fuse_head = !e->keep_all && C <= HEAD_FUSED_MAX_CLASSES;
    auto run_conv = [&](ConvLayer& L, bool head) {
      ConvArgs a{};
      const int H = h >> L.lvl, W = w >> L.lvl;
      a.in0 = sp(L.in); a.in1 = L.in2.C ? sp(L.in2) : nullptr; a.C0 = L.in.C; a.C1 = L.in2.C;
      a.wpk = L.wpk; a.scale = L.scale; a.bias = e->blob + L.b_off; a.out = sp(L.out);
      a.pool_out = L.do_pool ? sp(L.pool) : nullptr;
      a.N = nb; a.H = H; a.W = W; a.Cout = L.cout;
      const int TH = L.WAVES * L.MW;
      a.tiles_x = (W + 31) / 32; a.tiles_y = (H + TH - 1) / TH;
      a.nct = L.cout / (32 * L.NW); a.nchunks = L.nchunks;
      double px = (double)nb * H * W;
      double flops = 2.0 * px * L.cout * L.cin_real * 9;
      double bytes = px * P * 2.0 * (L.in.C + L.in2.C + (head ? 0 : L.cout)) + (L.do_pool ? px / 4 * P * 2.0 * L.cout : 0.0) + (double)L.cout * L.cin_real * 9 * 2.0 * P;
      if (head) {
        a.head_w = e->blob + e->head_w_off; a.head_b = e->blob + e->head_b_off; a.head_C = C;
        a.logits = lg; a.mask = mk; a.cable = cb; a.tape = tp;
        a.probs = pr; a.rule = outp->rule;
        a.t_cable = outp->t_cable; a.t_tape = outp->t_tape; a.bg_margin = outp->bg_margin; a.ct_margin = outp->ct_margin;
        flops += 2.0 * px * 32 * C;
        bytes += px * ((lg ? 4.0 * C : 0) + (pr ? 4.0 * C : 0) + (mk ? 1 : 0) + (cb ? 1 : 0) + (tp ? 1 : 0));
      }
      char lbl[112];
      snprintf(lbl, sizeof lbl, "%s%s|conv3x3_bias_relu_kernel<%d, %d, %d, %d, %d, %s, %s>", L.name.c_str(), head ? "+final+argmax" : "", P, L.KC, L.NW, L.MW, L.WAVES, L.do_pool ? "true" : "false", head ? "true" : "false");
      Lx.run(lbl, flops, bytes, [&] { return launch_conv(P, L, a, head, s); });
    };
    auto run_up = [&](int l, const Tensor& low) {
      const int H = h >> l, W = w >> l;
      double px = (double)nb * H * W;
      double bytes = px * P * 2.0 * low.C + px / 4 * P * 2.0 * low.C;
      // enough waves per LDS byte: the staged rows take 3*(W/2)*P*32 bytes per workgroup
      const int up_threads = (3 * (W / 2) * P * 32 > 32 * 1024) ? 1024 : (3 * (W / 2) * P * 32 > 12 * 1024 ? 512 : 256);
      char nm[64];
      snprintf(nm, sizeof nm, "up%d|upsample2x_kernel<%d>", l, P);
      Lx.run(nm, px * low.C * 8, bytes, [&] {
        if (P == 2) hipLaunchKernelGGL(upsample2x_kernel<2>, dim3((unsigned)(H / 2), (unsigned)(nb * (low.C / 16))), dim3(up_threads), 3 * (W / 2) * 2 * 32, s, sp(low), low.C, nb, H, W, sp(e->up[l]));
        else hipLaunchKernelGGL(upsample2x_kernel<1>, dim3((unsigned)(H / 2), (unsigned)(nb * (low.C / 16))), dim3(up_threads), 3 * (W / 2) * 1 * 32, s, sp(low), low.C, nb, H, W, sp(e->up[l]));
        return hipSuccess;
      });
    };
    size_t li = 0;
    for (int l = 0; l < 5; ++l) { run_conv(e->convs[li], false); ++li; run_conv(e->convs[li], false); ++li; }
    for (int l = 3; l >= 0; --l) {
      run_up(l, l == 3 ? e->x[4] : e->d[l + 1]);
      run_conv(e->convs[li], false); ++li;
      run_conv(e->convs[li], l == 0 && fuse_head); ++li;
    }
    // head as its own kernel only in debug mode (normally fused into conv0_4.conv2's epilogue)
    if (!fuse_head) {
      size_t total = (size_t)nb * hw;
      double bytes = (double)total * (P * 64 + (lg ? 4.0 * C : 0) + (mk ? 1 : 0) + (cb ? 1 : 0) + (tp ? 1 : 0));
      Lx.run(P == 2 ? "final+argmax|head_argmax_kernel<2>" : "final+argmax|head_argmax_kernel<1>", 2.0 * total * 32 * C, bytes, [&] {
        if (P == 2) hipLaunchKernelGGL(head_argmax_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, sp(e->d[0]), e->blob + e->head_w_off, e->blob + e->head_b_off, C, nb, h, w, lg, mk, cb, tp);
        else hipLaunchKernelGGL(head_argmax_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, sp(e->d[0]), e->blob + e->head_w_off, e->blob + e->head_b_off, C, nb, h, w, lg, mk, cb, tp);
        return hipSuccess;
      });
    }
    if (Lx.rc) return Lx.rc;
  }
  if (multi) {   // join: the caller's stream continues after every internal stream has drained
    for (int i = 0; i < e->nstreams; ++i) {
      HIP_TRY(e, hipEventRecord(e->ev_done[i], e->streams[i]));
      HIP_TRY(e, hipStreamWaitEvent(user_stream, e->ev_done[i], 0));
    }
  }
  return Lx.rc;
}

int unetpp_mask_stats(unetpp_engine* e, const uint8_t* dev_mask, int batch, int h, int w, uint32_t* dev_counts,
                      int32_t* dev_row_min, int32_t* dev_row_max, void* stream) {
  if (!e) return UNETPP_E_INVALID;
  if (!dev_mask || !dev_counts || !dev_row_min || !dev_row_max) return fail(e, UNETPP_E_INVALID, "null argument");
  if (batch < 1 || h < 1 || w < 1) return fail(e, UNETPP_E_INVALID, "bad shape %dx%dx%d", batch, h, w);
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const int C = e->cfg.num_classes;
  HIP_TRY(e, hipMemsetAsync(dev_counts, 0, (size_t)batch * C * sizeof(uint32_t), s));
  hipLaunchKernelGGL(mask_stats_kernel, dim3((unsigned)h, (unsigned)batch), dim3(256), 0, s, dev_mask, C, h, w,
                     (unsigned*)dev_counts, (int*)dev_row_min, (int*)dev_row_max);
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
  std::string nm(name);
  if (nm.size() == 4 && nm[0] == 'x' && nm[2] == '_') {
    int l = nm[1] - '0', j = nm[3] - '0';
    if (l >= 0 && l <= 4 && j == 0) t = &e->x[l];
    else if (l >= 0 && l <= 3 && j == 4 - l) t = &e->d[l];
  }
  if (!t) return fail(e, UNETPP_E_INVALID, "unknown tensor '%s'", name);
  if (t == &e->d[0] && !e->keep_all && e->cfg.num_classes <= HEAD_FUSED_MAX_CLASSES) return fail(e, UNETPP_E_STATE, "x0_4 is not materialised (head fused): call unetpp_debug_keep_intermediates(e, 1) before forward");
  HIP_TRY(e, hipSetDevice(e->cfg.device));
  int nb = e->last_b % e->mb == 0 ? std::min(e->mb, e->last_b) : e->last_b % e->mb;
  const int H = e->last_h >> t->lvl, W = e->last_w >> t->lvl;
  size_t total = (size_t)nb * t->C * H * W;
  if (total > max_floats) return fail(e, UNETPP_E_INVALID, "buffer too small: need %zu floats", total);
  float* tmp = nullptr;
  HIP_TRY(e, hipDeviceSynchronize());
  HIP_TRY(e, hipMalloc((void**)&tmp, total * sizeof(float)));
  if (e->P == 2) hipLaunchKernelGGL(unpack_nchw_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, (const half_t*)((const char*)t->p + e->last_slot_off), nb, t->C, H, W, tmp);
  else hipLaunchKernelGGL(unpack_nchw_kernel<1>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nullptr, (const half_t*)((const char*)t->p + e->last_slot_off), nb, t->C, H, W, tmp);
  hipError_t st = hipMemcpy(host_out, tmp, total * sizeof(float), hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (st != hipSuccess) return fail(e, UNETPP_E_HIP, "debug copy: %s", hipGetErrorString(st));
  return (long long)total;
}

}  // extern "C"
