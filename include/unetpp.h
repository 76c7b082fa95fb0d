/* unetpp.h — C ABI of the MI355X-native UNet++ (NestedUNet) inference engine.
 *
 * The reference (Chenxu1103/UNET-) is pure Python and has no FFI of its own; the boundary this
 * library replaces is the nn.Module protocol exercised by its frame loops.  Each entry point cites
 * the reference interface it stands in for (paths relative to the reference root).  The Python
 * drop-in (unet-_amd/nested_unet.py) binds exactly these symbols with ctypes; INTEGRATION.md shows
 * the stub a maintainer of the reference would add.
 *
 * Conventions: every function returns 0 on success or a negative UNETPP_E_* code; the message is
 * available from unetpp_last_error().  All pointers named dev_* are HIP device pointers on the
 * engine's device; `stream` is a hipStream_t passed as void* (NULL = the default stream).
 * unetpp_forward() is asynchronous on `stream` and never synchronises.  One engine per device;
 * an engine is not thread-safe, independent engines are.  The caller owns all I/O buffers; the
 * library owns packed weights and the activation workspace and never keeps a caller pointer.
 * Every call runs with the engine's device current and restores the calling thread's previous HIP device before
 * it returns.  One thread at a time per engine: the engine keeps per-call state (event pools, resize tables).
 *
 * Value range.  Activations live in HBM as fp16 planes (hi, or hi + lo in EXACT mode), so an activation (or a
 * float32 input value) beyond +-65504 is clamped and a NaN does not propagate the way it does in the fp32
 * reference (src/models/unetpp.py:23-26, src/models/simple_unet.py:94-128).  Neither happens silently: the
 * kernel that narrows such a value sets a sticky flag, see unetpp_status().
 */
#ifndef UNETPP_H
#define UNETPP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct unetpp_engine unetpp_engine;

enum {
  UNETPP_OK = 0,
  UNETPP_E_INVALID = -1,     /* bad argument (shape not a multiple of 16, batch too large, ...) */
  UNETPP_E_UNSUPPORTED = -2, /* e.g. pretrained_encoder=True (src/models/unetpp.py:52-65) */
  UNETPP_E_HIP = -3,         /* a HIP runtime call failed */
  UNETPP_E_STATE = -4        /* forward before weights were loaded, ... */
};

/* precision of the 3x3-conv stacks (the 1x1 head always runs in fp32):
 *   EXACT  fp16 MFMA with split operands (x = hi + lo for activations and weights, three MFMAs per
 *          product, fp32 accumulate): fp32-class accuracy; this is the parity-gated mode.
 *   FAST   plain fp16 operands, fp32 accumulate: single MFMA per product.
 *   EXACT8 the main term hi * hi of EXACT in fp16, its two small cross terms lo * w and x * w_lo from 8-bit
 *          operands (e5m2 activations, e4m3 weights) in ONE block-scaled K = 64 MFMA per tap pair
 *          (v_mfma_scale_f32_32x32x64_f8f6f4): 2/3 of EXACT's matrix-pipe cycles, logits within 1e-3 of
 *          the fp32 reference (measured <= 5e-4; EXACT: 1e-5, FAST: 5e-3).  Both architectures.   */
enum { UNETPP_PREC_EXACT = 0, UNETPP_PREC_FAST = 1, UNETPP_PREC_EXACT8 = 2 };

/* network architecture of an engine */
enum {
  UNETPP_ARCH_NESTED = 0,    /* NestedUNet, src/models/unetpp.py:28-135 (the north-star path) */
  UNETPP_ARCH_SIMPLE = 1     /* SimpleUNet, src/models/simple_unet.py:20-128: plain 4-level U-Net,
                                no BatchNorm, ConvTranspose2d(k=2,s=2) upsampling, cat([up, skip]); H, W multiples of 8 */
};

/* input formats accepted by unetpp_forward */
enum {
  UNETPP_IN_F32_NCHW = 0,    /* float32 [B,3,H,W] RGB in [0,1]: the tensor model(img_tensor) receives,
                                infer_two_stage_burr.py:292-295 */
  UNETPP_IN_U8_NHWC_BGR = 1  /* uint8 [B,H,W,3] BGR frame at model resolution: fuses the resize-free
                                part of preprocess_image (BGR->RGB, /255, HWC->CHW),
                                infer_two_stage_burr.py:122-127 */
};

typedef struct unetpp_config {
  int num_classes;   /* NestedUNet(num_classes=...)            src/models/unetpp.py:42 */
  int in_channels;   /* NestedUNet(input_channels=3)           src/models/unetpp.py:43 (only 3 supported) */
  int max_batch;     /* largest B passed to unetpp_forward */
  int max_h;         /* largest H (multiple of 16) */
  int max_w;         /* largest W (multiple of 16) */
  int precision;     /* UNETPP_PREC_* */
  int device;        /* HIP device ordinal: model.to(device)   infer_two_stage_burr.py:214 */
  int micro_batch;   /* frames pushed through the network per pass (0 = max_batch) */
  int streams;       /* passes in flight at once (1..4): each gets its own activation area and an internal
                        HIP stream, forked from / joined to the caller's stream with events, so HBM-bound
                        and MFMA-bound kernels of different passes overlap; 0 or 1 = serial */
  int arch;          /* UNETPP_ARCH_* */
} unetpp_config;

/* NestedUNet.__init__ + .to(device) (src/models/unetpp.py:40-91, infer_two_stage_burr.py:214):
 * allocates the activation workspace for (micro_batch, max_h, max_w) on cfg->device. */
int unetpp_create(const unetpp_config* cfg, unetpp_engine** out);

/* del model */
void unetpp_destroy(unetpp_engine* e);

/* Text of the last error on this engine (or, with e == NULL, of the last failed unetpp_create). */
const char* unetpp_last_error(const unetpp_engine* e);

const char* unetpp_version(void);

/* Size in bytes of the canonical weight blob for a (num_classes, in_channels) network:
 * 32-byte header {magic 'UNPP', version, num_classes, in_channels, n_layers, arch, 0,0} followed, for
 * each of the 18 3x3 convs in forward order (conv0_0.conv1, conv0_0.conv2, conv1_0.conv1, ...,
 * conv0_4.conv2) and then the 1x1 head, by the BN-folded fp32 weight in OIHW order and the
 * folded fp32 bias.  The host side (unet-_amd/packing.py) builds it from a state_dict. */
size_t unetpp_weights_blob_bytes(int num_classes, int in_channels);

/* Same for any architecture (header word 5 = arch).  SimpleUNet: the 8 encoder convs enc1.0 ... enc4.2, then
 * up3, up2, up1 (ConvTranspose2d weight in its native [Cin][Cout][2][2] order + bias), then dec3.0 ... dec1.2,
 * then the 1x1 head — the module's definition order (simple_unet.py:30-92); no BatchNorm to fold. */
size_t unetpp_weights_blob_bytes_arch(int arch, int num_classes, int in_channels);

/* model.load_state_dict(checkpoint['model'], strict=True) (infer_two_stage_burr.py:215-216):
 * takes the canonical blob from host memory, uploads it and repacks it on the device into the
 * kernels' layouts (per-channel power-of-two scaling, fp16 hi/lo planes, tile order). Synchronous. */
int unetpp_load_weights(unetpp_engine* e, const void* host_blob, size_t bytes);

/* Same, from a blob already in device memory (e.g. after the RCCL broadcast from rank 0);
 * asynchronous on `stream`. */
int unetpp_load_weights_device(unetpp_engine* e, const void* dev_blob, size_t bytes, void* stream);

/* outputs = model(img_tensor); probs = softmax(outputs,1); pred = argmax(probs,0).astype(uint8);
 * mask_cable = (pred==1); mask_tape = (pred==2)        (infer_two_stage_burr.py:294-304).
 *   dev_input   B frames in `in_format`
 *   dev_logits  float32 [B,num_classes,H,W] or NULL      (NestedUNet.forward return, unetpp.py:119,135)
 *   dev_mask    uint8 [B,H,W] class index (first maximal class) or NULL
 *   dev_cable / dev_tape  uint8 [B,H,W] 0/1 masks of class 1 / class 2, or NULL
 * H and W must be multiples of 16 (the reference raises in torch.cat otherwise), B <= max_batch. */
int unetpp_forward(unetpp_engine* e, const void* dev_input, int in_format, int batch, int h, int w,
                   float* dev_logits, uint8_t* dev_mask, uint8_t* dev_cable, uint8_t* dev_tape,
                   void* stream);

/* ---- value-range status -------------------------------------------------------------------------
 * Sticky flags set by the kernels of any forward since creation (or since the last clearing read):
 *   UNETPP_STATUS_OVERFLOW  a conv / transposed-conv output or a float32 input value exceeded the fp16
 *                           range and was clamped to +-65504: results differ from the fp32 reference
 *   UNETPP_STATUS_NAN       a NaN in a float32 input, or a non-finite weight / bias in a loaded blob (the only ways
 *                           a NaN can reach an accumulator: fp16 operands cannot overflow fp32 sums); the reference
 *                           would carry it to its logits, this engine replaces it (clamp -> +-65504, ReLU -> 0)
 * unetpp_status synchronises the device (hipDeviceSynchronize), copies the flags to *flags and, with
 * clear != 0, resets them.  0 = every value since the last clear was representable. */
enum { UNETPP_STATUS_OVERFLOW = 1, UNETPP_STATUS_NAN = 2 };
int unetpp_status(unetpp_engine* e, uint32_t* flags, int clear);

/* ---- probability outputs and thresholded class rules -------------------------
 * Half of the reference's frame loops do not take a plain argmax: they compute
 * probs = softmax_np(outputs[0].transpose(1,2,0)) on the host and derive the cable / tape masks with
 * a thresholded rule.  unetpp_forward_ex runs the softmax (fp32) and the rule in the same epilogue as
 * the 1x1 head, so neither logits nor probabilities need to leave the GPU. */
enum {
  UNETPP_RULE_ARGMAX = 0,        /* (pred==1), (pred==2)                infer_two_stage_burr.py:303-304 */
  UNETPP_RULE_THRESHOLDED = 1,   /* thresholded_argmax(probs, t_cable, t_tape, bg_margin)
                                    infer_video_3class_best.py:56-83, infer_video_strict.py:36-63 */
  UNETPP_RULE_STRICT_BG = 2,     /* strict_threshold_with_bg_check(probs, t_cable, t_tape, bg_margin)
                                    infer_video_fixed.py:35-83 */
  UNETPP_RULE_EXCLUSIVE = 3      /* exclusive_threshold(probs, t_cable, t_tape, bg_margin, ct_margin)
                                    infer_video_robust.py:70-99 */
};

typedef struct unetpp_outputs {
  float* dev_logits;     /* float32 [B,C,H,W] or NULL */
  float* dev_probs;      /* float32 [B,C,H,W] softmax over C, or NULL (the reference builds HxWxC on the host) */
  uint8_t* dev_mask;     /* uint8 [B,H,W] plain argmax class index, or NULL */
  uint8_t* dev_cable;    /* uint8 [B,H,W] 0/1 under `rule`, or NULL */
  uint8_t* dev_tape;     /* uint8 [B,H,W] 0/1 under `rule`, or NULL */
  int rule;              /* UNETPP_RULE_* (rules 1-3 need num_classes >= 3: bg, cable, tape = classes 0,1,2) */
  float t_cable, t_tape, bg_margin, ct_margin;
} unetpp_outputs;

int unetpp_forward_ex(unetpp_engine* e, const void* dev_input, int in_format, int batch, int h, int w,
                      const unetpp_outputs* out, void* stream);

/* ---- mask statistics on the device -------------------------------------------
 * From a uint8 class-index mask [B,H,W] (e.g. the dev_mask of a forward): per-frame class pixel counts
 * (np.sum(mask_cable) / coverage, infer_two_stage_burr.py:333-340, src/utils/geometry_enhanced.py:151-152) and,
 * per class and row, the first and last column of that class — the operands of _compute_width_per_row
 * (geometry_enhanced.py:45-74: width = xs.max() - xs.min() + 1).  The reference's smoothing and
 * connected-component filtering (cv2) stay on the host.
 *   dev_counts   uint32 [B,num_classes]      (zeroed by this call)
 *   dev_row_min  int32  [B,num_classes,H]    W  when the row has no pixel of the class
 *   dev_row_max  int32  [B,num_classes,H]    -1 when the row has no pixel of the class
 * Asynchronous on `stream`. */
int unetpp_mask_stats(unetpp_engine* e, const uint8_t* dev_mask, int batch, int h, int w, uint32_t* dev_counts,
                      int32_t* dev_row_min, int32_t* dev_row_max, void* stream);

/* ---- frame glue either side of the model --------------------------------
 * unetpp_resize_linear_u8 replaces `cv2.resize(frame_rgb, target_size, interpolation=cv2.INTER_LINEAR)`
 * of preprocess_image (infer_two_stage_burr.py:124; also the --normalize-resolution resize, :281):
 *   dev_src uint8 [B,src_h,src_w,channels] interleaved (channels 1..4)  ->  dev_dst uint8 [B,dst_h,dst_w,channels]
 * OpenCV's published fixed-point algorithm (11-bit coefficients; see oracle/unetpp_oracle.py
 * cv2_resize_linear_u8_np — parity unpinned: cv2 is not installable in the build environment).
 * The result feeds unetpp_forward(_ex) with UNETPP_IN_U8_NHWC_BGR, which does BGR->RGB, /255 and the layout change.
 *
 * unetpp_resize_nearest_roi_u8 replaces infer_two_stage_burr.py:303-314 for one class:
 *   (pred == match_class).astype(uint8)  [match_class < 0: the mask itself]
 *   -> cv2.resize(mask, (dst_w, dst_h), interpolation=cv2.INTER_NEAREST)
 *   -> zeros outside rows [y1, y2) x columns [x1, x2) (Python slice semantics for non-negative bounds; pass
 *      0, 0, dst_w, dst_h for "no ROI").
 *   dev_src uint8 [B,src_h,src_w]  ->  dev_dst uint8 [B,dst_h,dst_w]
 * Both are asynchronous on `stream`, except that the first call for a new (source, destination) extent builds
 * the index tables on the host and uploads them with a blocking copy. */
int unetpp_resize_linear_u8(unetpp_engine* e, const uint8_t* dev_src, int batch, int src_h, int src_w, int channels,
                            uint8_t* dev_dst, int dst_h, int dst_w, void* stream);
int unetpp_resize_nearest_roi_u8(unetpp_engine* e, const uint8_t* dev_src, int batch, int src_h, int src_w,
                                 int match_class, uint8_t* dev_dst, int dst_h, int dst_w, int x1, int y1, int x2,
                                 int y2, void* stream);

/* Bytes of device memory held by the engine (workspace + packed weights). */
size_t unetpp_workspace_bytes(const unetpp_engine* e);

/* ---- measurement hooks (bench.py roofline leg) ------------------------------------------------
 * With profiling on, every kernel launch of the next forward is bracketed by HIP events recorded
 * on the launch stream.  unetpp_profile_read synchronises those events and returns, per launch in
 * issue order, its duration in milliseconds; unetpp_profile_name gives the launch's label. */
int unetpp_profile_enable(unetpp_engine* e, int on);
int unetpp_profile_count(const unetpp_engine* e);
int unetpp_profile_read(unetpp_engine* e, float* ms_out, int n);
const char* unetpp_profile_name(const unetpp_engine* e, int i);
/* algorithmic FLOPs and minimum HBM bytes (read+write) of launch i for the last forward's shape */
int unetpp_profile_work(const unetpp_engine* e, int i, double* flops, double* bytes);

/* ---- debug (layer-by-layer parity tests) -------------------------------------------------------
 * Copies the named activation of the LAST micro-batch processed ("x0_0", "x1_0", ..., "x0_4"; also a block's first conv
 * "x1_0a" and a pooled tensor "x1_0p" [b,C,h/2,w/2]) to host memory as float32 [b,C,h,w]; returns the number of floats
 * written or a negative error.  "name#hi", "name#lo", "name#x8" give one stored plane instead of the reconstructed value
 * (fp16 hi; fp16 lo, or EXACT8's decoded e5m2(2^8 lo); EXACT8's decoded e5m2(2^-3 v)). */
long long unetpp_debug_read(unetpp_engine* e, const char* name, float* host_out, size_t max_floats);

/* x0_4 is normally never written to HBM (the 1x1 head + argmax run in the epilogue of conv0_4.conv2).
 * With on != 0 the next forwards materialise it and run the head as a separate kernel, so that
 * unetpp_debug_read("x0_4") works. */
int unetpp_debug_keep_intermediates(unetpp_engine* e, int on);

#ifdef __cplusplus
}
#endif
#endif /* UNETPP_H */
