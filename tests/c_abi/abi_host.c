/* abi_host.c — a host written in plain C99 against include/unetpp.h only (no Python, no torch, no C++):
 * what a non-Python caller of the drop-in boundary looks like.  Test infrastructure (tests/test_c_abi.py):
 *
 *   abi_host BLOB FRAMES_U8 OUT  num_classes batch h w precision
 *
 * BLOB       canonical weight blob (unet_amd.packing.build_blob)
 * FRAMES_U8  uint8 [B,H,W,3] BGR frames at model resolution
 * OUT        written: float32 logits [B,C,H,W], then uint8 mask [B,H,W], then uint8 cable, uint8 tape
 *
 * Device memory comes from the HIP runtime's C API, resolved with dlopen so that this file needs no HIP
 * headers (the ABI itself only ever sees void* device pointers). */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "unetpp.h"

typedef int (*hip_malloc_t)(void**, size_t);
typedef int (*hip_free_t)(void*);
typedef int (*hip_memcpy_t)(void*, const void*, size_t, int);
typedef int (*hip_sync_t)(void);
enum { H2D = 1, D2H = 2 };

static void* slurp(const char* path, size_t* n) {
  FILE* f = fopen(path, "rb");
  if (!f) { perror(path); exit(2); }
  fseek(f, 0, SEEK_END);
  *n = (size_t)ftell(f);
  fseek(f, 0, SEEK_SET);
  void* p = malloc(*n);
  if (!p || fread(p, 1, *n, f) != *n) { fprintf(stderr, "short read: %s\n", path); exit(2); }
  fclose(f);
  return p;
}

#define CHECK(call)                                                                          \
  do {                                                                                       \
    int rc_ = (call);                                                                        \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, unetpp_last_error(e)); return 1; } \
  } while (0)

int main(int argc, char** argv) {
  if (argc != 9) { fprintf(stderr, "usage: %s BLOB FRAMES OUT classes batch h w precision\n", argv[0]); return 2; }
  const int C = atoi(argv[4]), B = atoi(argv[5]), H = atoi(argv[6]), W = atoi(argv[7]), prec = atoi(argv[8]);
  void* hip = dlopen("libamdhip64.so", RTLD_NOW | RTLD_GLOBAL);
  if (!hip) { fprintf(stderr, "dlopen libamdhip64.so: %s\n", dlerror()); return 2; }
  hip_malloc_t hipMalloc; hip_free_t hipFree; hip_memcpy_t hipMemcpy; hip_sync_t hipDeviceSynchronize;
  *(void**)(&hipMalloc) = dlsym(hip, "hipMalloc");            /* the POSIX-sanctioned way to take a function from dlsym */
  *(void**)(&hipFree) = dlsym(hip, "hipFree");
  *(void**)(&hipMemcpy) = dlsym(hip, "hipMemcpy");
  *(void**)(&hipDeviceSynchronize) = dlsym(hip, "hipDeviceSynchronize");
  if (!hipMalloc || !hipFree || !hipMemcpy || !hipDeviceSynchronize) { fprintf(stderr, "HIP symbols missing\n"); return 2; }

  size_t blob_bytes, frame_bytes;
  void* blob = slurp(argv[1], &blob_bytes);
  void* frames = slurp(argv[2], &frame_bytes);
  const size_t px = (size_t)B * H * W;
  if (frame_bytes != px * 3) { fprintf(stderr, "frames file has %zu bytes, expected %zu\n", frame_bytes, px * 3); return 2; }
  if (blob_bytes != unetpp_weights_blob_bytes(C, 3)) { fprintf(stderr, "blob size mismatch\n"); return 2; }

  unetpp_engine* e = NULL;
  unetpp_config cfg = {0};
  cfg.num_classes = C; cfg.in_channels = 3; cfg.max_batch = B; cfg.max_h = H; cfg.max_w = W;
  cfg.precision = prec; cfg.device = 0; cfg.micro_batch = 0; cfg.streams = 1; cfg.arch = UNETPP_ARCH_NESTED;
  CHECK(unetpp_create(&cfg, &e));
  CHECK(unetpp_load_weights(e, blob, blob_bytes));

  void *d_in = NULL, *d_logits = NULL, *d_mask = NULL, *d_cable = NULL, *d_tape = NULL;
  if (hipMalloc(&d_in, px * 3) || hipMalloc(&d_logits, px * C * sizeof(float)) || hipMalloc(&d_mask, px) ||
      hipMalloc(&d_cable, px) || hipMalloc(&d_tape, px)) { fprintf(stderr, "hipMalloc failed\n"); return 1; }
  if (hipMemcpy(d_in, frames, px * 3, H2D)) { fprintf(stderr, "hipMemcpy H2D failed\n"); return 1; }
  CHECK(unetpp_forward(e, d_in, UNETPP_IN_U8_NHWC_BGR, B, H, W, (float*)d_logits, (uint8_t*)d_mask, (uint8_t*)d_cable,
                       (uint8_t*)d_tape, NULL));
  if (hipDeviceSynchronize()) { fprintf(stderr, "hipDeviceSynchronize failed\n"); return 1; }

  /* error path: H not a multiple of 16 must be refused with a message, like the reference's torch.cat failure */
  if (unetpp_forward(e, d_in, UNETPP_IN_U8_NHWC_BGR, 1, H - 8, W, NULL, (uint8_t*)d_mask, NULL, NULL, NULL) == 0 ||
      unetpp_last_error(e)[0] == 0) { fprintf(stderr, "bad shape was accepted\n"); return 1; }

  /* every value of this run fitted the fp16 activation planes (unetpp_status synchronises and reads the sticky flags) */
  uint32_t flags = 99;
  CHECK(unetpp_status(e, &flags, 1));
  if (flags != 0) { fprintf(stderr, "range status %u after an ordinary forward\n", (unsigned)flags); return 1; }

  float* logits = (float*)malloc(px * C * sizeof(float));
  uint8_t* bytes = (uint8_t*)malloc(px * 3);
  if (hipMemcpy(logits, d_logits, px * C * sizeof(float), D2H) || hipMemcpy(bytes, d_mask, px, D2H) ||
      hipMemcpy(bytes + px, d_cable, px, D2H) || hipMemcpy(bytes + 2 * px, d_tape, px, D2H)) { fprintf(stderr, "D2H failed\n"); return 1; }
  FILE* f = fopen(argv[3], "wb");
  if (!f || fwrite(logits, sizeof(float), px * C, f) != px * C || fwrite(bytes, 1, px * 3, f) != px * 3) { perror(argv[3]); return 1; }
  fclose(f);
  hipFree(d_in); hipFree(d_logits); hipFree(d_mask); hipFree(d_cable); hipFree(d_tape);
  printf("%s: ok, workspace %zu bytes\n", unetpp_version(), unetpp_workspace_bytes(e));
  unetpp_destroy(e);
  return 0;
}
