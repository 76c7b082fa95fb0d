"""ORACLE — TEST INFRASTRUCTURE ONLY.  CPU restatement of the reference UNet++ inference hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product (unet-_amd/) never does and fails loudly when its HIP library is missing.

What is restated (citations are /root/reference paths):
  * ConvBlock.forward           src/models/unetpp.py:23-26   relu(bn1(conv1(x))), relu(bn2(conv2(x)))
  * NestedUNet.forward (eval)   src/models/unetpp.py:104-119 encoder column + outer decoder diagonal
  * nn.MaxPool2d(2,2)           src/models/unetpp.py:75
  * nn.Upsample(2,'bilinear',align_corners=True)  src/models/unetpp.py:76
  * torch.cat([skip, up],1)     src/models/unetpp.py:112-116  (skip channels first)
  * final 1x1 conv              src/models/unetpp.py:85,119
  * softmax -> argmax -> uint8, class masks   infer_two_stage_burr.py:299-304
  *  SimpleUNet.forward     src/models/simple_unet.py:94-128 (ConvTranspose2d k2 s2, cat([up, skip]))
  *  per-row mask widths     src/utils/geometry_enhanced.py:45-74
  *  softmax probabilities + thresholded / strict / exclusive class rules
                                infer_video_3class_best.py:50-83, infer_video_strict.py:36-63,
                                infer_video_fixed.py:35-83, infer_video_robust.py:64-99

The arithmetic of conv / batch-norm / bilinear upsampling lives in a third-party dependency of the
reference, PyTorch (requirements.txt:1 pins only torch>=1.10.0; the container has 2.10.0+rocm7.0).
Two restatements of its published algorithms are given:
  numpy_forward   plain NumPy (float32, or float64 for a high-precision cross-check)
  torch_forward   the same graph through torch.nn.functional CPU ops (what the reference's
                  --device cpu path actually executes); this one is timed as the cpu_baseline.

PARITY PIN: the reference has no tests, golden vectors or fixtures for this path, so
the oracle is pinned by outputs of the reference model itself, imported in the build container by
oracle/make_golden.py and committed under tests/golden/ (tests/test_oracle_golden.py checks both
restatements against them).
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5  # nn.BatchNorm2d default, unetpp.py:18,20

BLOCKS = ("conv0_0", "conv1_0", "conv2_0", "conv3_0", "conv4_0", "conv3_1", "conv2_2", "conv1_3", "conv0_4")


# ----------------------------------------------------------------------------- NumPy restatement
def conv3x3_np(x, w, b):
    """nn.Conv2d(k=3, padding=1, stride=1, bias=True) as cross-correlation (unetpp.py:17,19).
    x [B,Ci,H,W], w [Co,Ci,3,3], b [Co] -> [B,Co,H,W]; accumulates in x.dtype."""
    B, Ci, H, W = x.shape
    Co = w.shape[0]
    xp = np.zeros((B, Ci, H + 2, W + 2), dtype=x.dtype)
    xp[:, :, 1:-1, 1:-1] = x
    out = np.empty((B, Co, H, W), dtype=x.dtype)
    for n in range(B):
        acc = np.zeros((Co, H * W), dtype=x.dtype)
        for dy in range(3):
            for dx in range(3):
                patch = np.ascontiguousarray(xp[n, :, dy:dy + H, dx:dx + W]).reshape(Ci, H * W)
                acc += w[:, :, dy, dx].astype(x.dtype) @ patch
        out[n] = (acc + b.astype(x.dtype)[:, None]).reshape(Co, H, W)
    return out


def conv1x1_np(x, w, b):
    """nn.Conv2d(32, C, kernel_size=1) (unetpp.py:85)."""
    B, Ci, H, W = x.shape
    wm = w.reshape(w.shape[0], Ci).astype(x.dtype)
    out = np.einsum("oc,bchw->bohw", wm, x, optimize=True)
    return out + b.astype(x.dtype)[None, :, None, None]


def batchnorm_eval_np(x, gamma, beta, mean, var):
    """nn.BatchNorm2d in eval mode: (x-mean)/sqrt(var+eps)*gamma+beta."""
    dt = x.dtype
    inv = (1.0 / np.sqrt(var.astype(dt) + dt.type(BN_EPS))).astype(dt)
    return (x - mean.astype(dt)[None, :, None, None]) * (inv * gamma.astype(dt))[None, :, None, None] \
        + beta.astype(dt)[None, :, None, None]


def relu_np(x):
    return np.maximum(x, 0)


def maxpool2x2_np(x):
    """nn.MaxPool2d(kernel_size=2, stride=2): floor mode, no padding (unetpp.py:75)."""
    B, C, H, W = x.shape
    h2, w2 = H // 2, W // 2
    v = x[:, :, :h2 * 2, :w2 * 2].reshape(B, C, h2, 2, w2, 2)
    return v.max(axis=(3, 5))


def bilinear_axis_tables(n_in: int, n_out: int):
    """Source indices and weights for align_corners=True (ATen area_pixel_compute_scale /
    compute_source_index_and_lambda): scale=(in-1)/(out-1) in float32; src=scale*dst;
    i0=int(src); i1=i0+(i0<in-1); l1=src-i0; l0=1-l1."""
    scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    src = scale * np.arange(n_out, dtype=np.float32)
    i0 = src.astype(np.int64)
    i1 = i0 + (i0 < n_in - 1)
    l1 = np.clip(src - i0.astype(np.float32), 0, 1).astype(np.float32)
    l0 = (np.float32(1) - l1).astype(np.float32)
    return i0, i1, l0, l1


def upsample2x_bilinear_ac_np(x):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) (unetpp.py:76).
    Interpolates along x inside each source row, then between the two rows."""
    B, C, H, W = x.shape
    dt = x.dtype
    y0, y1, ly0, ly1 = bilinear_axis_tables(H, 2 * H)
    x0, x1, lx0, lx1 = bilinear_axis_tables(W, 2 * W)
    lx0 = lx0.astype(dt)[None, None, None, :]; lx1 = lx1.astype(dt)[None, None, None, :]
    ly0 = ly0.astype(dt)[None, None, :, None]; ly1 = ly1.astype(dt)[None, None, :, None]
    top = x[:, :, y0][:, :, :, x0] * lx0 + x[:, :, y0][:, :, :, x1] * lx1
    bot = x[:, :, y1][:, :, :, x0] * lx0 + x[:, :, y1][:, :, :, x1] * lx1
    return top * ly0 + bot * ly1


def conv_block_np(x, sd, name):
    """ConvBlock.forward (unetpp.py:23-26)."""
    for j in (1, 2):
        x = conv3x3_np(x, sd[f"{name}.conv{j}.weight"], sd[f"{name}.conv{j}.bias"])
        x = batchnorm_eval_np(x, sd[f"{name}.bn{j}.weight"], sd[f"{name}.bn{j}.bias"],
                              sd[f"{name}.bn{j}.running_mean"], sd[f"{name}.bn{j}.running_var"])
        x = relu_np(x)
    return x


def numpy_forward(sd: dict, x: np.ndarray, dtype=np.float32, return_intermediates: bool = False):
    """NestedUNet.forward in eval mode (unetpp.py:104-119). x [B,3,H,W] in [0,1] -> logits [B,C,H,W]."""
    if x.shape[2] % 16 or x.shape[3] % 16:
        # the reference raises inside torch.cat for such sizes (SURVEY.md §7 hard part 7)
        raise RuntimeError("Sizes of tensors must match: H and W must be multiples of 16")
    x = x.astype(dtype)
    t = {}
    t["x0_0"] = conv_block_np(x, sd, "conv0_0")
    t["x1_0"] = conv_block_np(maxpool2x2_np(t["x0_0"]), sd, "conv1_0")
    t["x2_0"] = conv_block_np(maxpool2x2_np(t["x1_0"]), sd, "conv2_0")
    t["x3_0"] = conv_block_np(maxpool2x2_np(t["x2_0"]), sd, "conv3_0")
    t["x4_0"] = conv_block_np(maxpool2x2_np(t["x3_0"]), sd, "conv4_0")
    cat = np.concatenate
    t["x3_1"] = conv_block_np(cat([t["x3_0"], upsample2x_bilinear_ac_np(t["x4_0"])], 1), sd, "conv3_1")
    t["x2_2"] = conv_block_np(cat([t["x2_0"], upsample2x_bilinear_ac_np(t["x3_1"])], 1), sd, "conv2_2")
    t["x1_3"] = conv_block_np(cat([t["x1_0"], upsample2x_bilinear_ac_np(t["x2_2"])], 1), sd, "conv1_3")
    t["x0_4"] = conv_block_np(cat([t["x0_0"], upsample2x_bilinear_ac_np(t["x1_3"])], 1), sd, "conv0_4")
    logits = conv1x1_np(t["x0_4"], sd["final.weight"], sd["final.bias"])
    if return_intermediates:
        t["logits"] = logits
        return logits, t
    return logits


def softmax_np(logits, axis=1):
    m = logits.max(axis=axis, keepdims=True)
    e = np.exp(logits - m)
    return e / e.sum(axis=axis, keepdims=True)


def masks_from_logits(logits: np.ndarray):
    """infer_two_stage_burr.py:299-304: softmax(dim=1) -> np.argmax (first max index) -> uint8;
    mask_cable=(pred==1), mask_tape=(pred==2) as uint8.  softmax is monotone, so argmax of the
    probabilities equals argmax of the logits except where exp() rounds two different logits to the
    same probability; the probabilities are used here, as the reference does."""
    probs = softmax_np(logits.astype(np.float32), axis=1)
    pred = np.argmax(probs, axis=1).astype(np.uint8)
    return pred, (pred == 1).astype(np.uint8), (pred == 2).astype(np.uint8)


def top2_margin(logits: np.ndarray) -> np.ndarray:
    """Per-pixel gap between the largest and second-largest logit (for margin-aware flip accounting)."""
    if logits.shape[1] < 2:                     # a single class cannot flip
        return np.full(logits.shape[:1] + logits.shape[2:], np.inf, np.float32)
    s = np.sort(logits, axis=1)
    return (s[:, -1] - s[:, -2]).astype(np.float32)


# ----------------------------------------------------------------------------- SimpleUNet
def conv_transpose2x2_np(x, w, b):
    """nn.ConvTranspose2d(Cin, Cout, kernel_size=2, stride=2) (src/models/simple_unet.py:62-64):
    out[n,co,2y+dy,2x+dx] = b[co] + sum_ci x[n,ci,y,x] * w[ci,co,dy,dx]   (weight layout [Cin,Cout,2,2])."""
    B, Ci, H, W = x.shape
    Co = w.shape[1]
    out = np.empty((B, Co, 2 * H, 2 * W), dtype=x.dtype)
    for dy in range(2):
        for dx in range(2):
            out[:, :, dy::2, dx::2] = np.einsum("bchw,co->bohw", x, w[:, :, dy, dx].astype(x.dtype), optimize=True)
    return out + b.astype(x.dtype)[None, :, None, None]


def simple_unet_numpy_forward(sd: dict, x: np.ndarray, dtype=np.float32, return_intermediates: bool = False):
    """SimpleUNet.forward (src/models/simple_unet.py:94-128): four Conv-ReLU-Conv-ReLU encoders with 2x2
    max-pools, three ConvTranspose2d upsamplers, cat([up, skip]) decoders, 1x1 head."""
    if x.shape[2] % 8 or x.shape[3] % 8:
        raise RuntimeError("Sizes of tensors must match: H and W must be multiples of 8")
    x = x.astype(dtype)

    def cr2(x, name):
        for j in (0, 2):
            x = relu_np(conv3x3_np(x, sd[f"{name}.{j}.weight"], sd[f"{name}.{j}.bias"]))
        return x

    t = {}
    t["enc1"] = cr2(x, "enc1")
    t["enc2"] = cr2(maxpool2x2_np(t["enc1"]), "enc2")
    t["enc3"] = cr2(maxpool2x2_np(t["enc2"]), "enc3")
    t["enc4"] = cr2(maxpool2x2_np(t["enc3"]), "enc4")
    up3 = conv_transpose2x2_np(t["enc4"], sd["up3.weight"], sd["up3.bias"])
    t["dec3"] = cr2(np.concatenate([up3, t["enc3"]], 1), "dec3")
    up2 = conv_transpose2x2_np(t["dec3"], sd["up2.weight"], sd["up2.bias"])
    t["dec2"] = cr2(np.concatenate([up2, t["enc2"]], 1), "dec2")
    up1 = conv_transpose2x2_np(t["dec2"], sd["up1.weight"], sd["up1.bias"])
    t["dec1"] = cr2(np.concatenate([up1, t["enc1"]], 1), "dec1")
    logits = conv1x1_np(t["dec1"], sd["final.weight"], sd["final.bias"])
    if return_intermediates:
        return logits, t
    return logits


def simple_unet_torch_forward(sd: dict, x, return_intermediates: bool = False):
    """The same graph through torch.nn.functional CPU ops (what the reference dispatches to)."""
    import torch
    import torch.nn.functional as F

    def T(a):
        return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))

    x = T(x).float()
    if x.shape[2] % 8 or x.shape[3] % 8:
        raise RuntimeError("Sizes of tensors must match: H and W must be multiples of 8")

    def cr2(x, name):
        for j in (0, 2):
            x = F.relu(F.conv2d(x, T(sd[f"{name}.{j}.weight"]), T(sd[f"{name}.{j}.bias"]), padding=1))
        return x

    with torch.no_grad():
        t = {}
        t["enc1"] = cr2(x, "enc1")
        t["enc2"] = cr2(F.max_pool2d(t["enc1"], 2, 2), "enc2")
        t["enc3"] = cr2(F.max_pool2d(t["enc2"], 2, 2), "enc3")
        t["enc4"] = cr2(F.max_pool2d(t["enc3"], 2, 2), "enc4")
        prev = t["enc4"]
        for l in (3, 2, 1):
            up = F.conv_transpose2d(prev, T(sd[f"up{l}.weight"]), T(sd[f"up{l}.bias"]), stride=2)
            prev = t[f"dec{l}"] = cr2(torch.cat([up, t[f"enc{l}"]], 1), f"dec{l}")
        logits = F.conv2d(prev, T(sd["final.weight"]), T(sd["final.bias"]))
    if return_intermediates:
        return logits.numpy(), {k: v.numpy() for k, v in t.items()}
    return logits.numpy()


# ----------------------------------------------------------------------------- probability rules
def softmax_last_np(x):
    """softmax_np of the thresholded frame loops (infer_video_3class_best.py:50-53, infer_video_robust.py:64-67):
    exp(x - max) / sum over the LAST axis of an HxWxC float32 array."""
    e_x = np.exp(x - np.max(x, axis=-1, keepdims=True))
    return e_x / np.sum(e_x, axis=-1, keepdims=True)


def thresholded_argmax_np(probs, t_cable=0.45, t_tape=0.50, bg_margin=0.15):
    """infer_video_3class_best.py:56-83 (infer_video_strict.py:36-63 is the same rule with defaults 0.60/0.65/0.30)."""
    bg, cable, tape = probs[..., 0], probs[..., 1], probs[..., 2]
    winner = np.argmax(probs[..., :3], axis=-1)
    mask_cable = (winner == 1) & (cable >= t_cable) & ((cable - bg) >= bg_margin)
    mask_tape = (winner == 2) & (tape >= t_tape) & ((tape - bg) >= bg_margin)
    return mask_cable.astype(np.uint8), mask_tape.astype(np.uint8)


def strict_threshold_with_bg_check_np(probs, t_cable=0.6, t_tape=0.65, bg_margin=0.4):
    """infer_video_fixed.py:35-83: winner + confidence + background-probability ceiling (the overlap branch is
    unreachable because `winner` is exclusive)."""
    bg, cable, tape = probs[..., 0], probs[..., 1], probs[..., 2]
    winner = np.argmax(probs[..., :3], axis=-1)
    mask_cable = (winner == 1) & (cable >= t_cable) & (bg <= bg_margin)
    mask_tape = (winner == 2) & (tape >= t_tape) & (bg <= bg_margin)
    return mask_cable.astype(np.uint8), mask_tape.astype(np.uint8)


def exclusive_threshold_np(probs, t_cable=0.55, t_tape=0.60, bg_margin=0.20, ct_margin=0.10):
    """infer_video_robust.py:70-99: candidates by confidence and margin over background, then the stronger of
    cable/tape by ct_margin; remaining overlap (only possible for ct_margin <= 0) goes to the larger probability."""
    pbg, pc, pt = probs[..., 0], probs[..., 1], probs[..., 2]
    cand_c = (pc >= t_cable) & (pc >= pbg + bg_margin)
    cand_t = (pt >= t_tape) & (pt >= pbg + bg_margin)
    cable = cand_c & (pc >= pt + ct_margin)
    tape = cand_t & (pt >= pc + ct_margin)
    overlap = cable & tape
    if np.any(overlap):
        cable[overlap] = pc[overlap] >= pt[overlap]
        tape[overlap] = ~cable[overlap]
    return cable.astype(np.uint8), tape.astype(np.uint8)


def width_per_row_np(mask):
    """_compute_width_per_row(mask, smooth=False) of src/utils/geometry_enhanced.py:45-74: per row,
    xs.max() - xs.min() + 1 over the non-zero columns, 0 for an empty row (float32)."""
    H, W = mask.shape
    widths = np.zeros(H, dtype=np.float32)
    for y in range(H):
        xs = np.where(mask[y] > 0)[0]
        if xs.size > 0:
            widths[y] = float(xs.max() - xs.min() + 1)
    return widths


def mask_stats_np(pred, num_classes):
    """counts[b,c] = np.sum(pred[b]==c) (infer_two_stage_burr.py:333-334) and widths[b,c,:] as above."""
    B = pred.shape[0]
    counts = np.stack([np.bincount(pred[b].ravel(), minlength=num_classes)[:num_classes] for b in range(B)]).astype(np.int64)
    widths = np.stack([np.stack([width_per_row_np((pred[b] == c).astype(np.uint8)) for c in range(num_classes)]) for b in range(B)])
    return counts, widths


RULES = {"thresholded_argmax": thresholded_argmax_np, "strict_bg_check": strict_threshold_with_bg_check_np,
         "exclusive": exclusive_threshold_np}


def rule_masks_from_logits(logits_nchw, rule, **params):
    """Frame-loop tail of the thresholded scripts for a batch: probs = softmax_np(outputs[i].transpose(1,2,0))
    (infer_video_3class_best.py:197), then the rule.  Returns (cable, tape, probs[B,H,W,C])."""
    hwc = np.transpose(logits_nchw.astype(np.float32), (0, 2, 3, 1))
    probs = softmax_last_np(hwc)
    c, t = RULES[rule](probs, **params)
    return c, t, probs


# ----------------------------------------------------------------------------- frame glue
# PARITY UNPINNED for the two cv2.resize restatements below: cv2 (opencv-python; the reference pins no version,
# requirements.txt) is not installed here and the reference holds no resized fixture, so they restate the
# algorithm OpenCV 4.x publishes in modules/imgproc/src/resize.cpp (generic, non-IPP path) and are only
# cross-checked against torch's float interpolation (tests/test_oracle_golden.py).  map_roi_to_original and the
# ROI clip are pinned by the reference's own function (tests/golden/roi_map.json).
INTER_RESIZE_COEF_BITS = 11
INTER_RESIZE_COEF_SCALE = 1 << INTER_RESIZE_COEF_BITS


def cv2_linear_tables(n_src: int, n_dst: int):
    """Source index and the two 11-bit fixed-point coefficients per destination index, as resizeGeneric_ builds
    them for INTER_LINEAR: fx = (float)((d + 0.5) * scale - 0.5) with scale = 1 / (n_dst / n_src) in double,
    s = floor(fx), clamped at both borders with fx = 0, coefficients saturate_cast<short>(c * 2048) with
    round-half-to-even."""
    inv_scale = np.float64(n_dst) / np.float64(n_src)
    scale = np.float64(1.0) / inv_scale
    d = np.arange(n_dst, dtype=np.float64)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int32)
    fx = (fx - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    fx[lo] = 0.0; s[lo] = 0
    hi = s >= n_src - 1
    fx[hi] = 0.0; s[hi] = n_src - 1
    c0 = (np.float32(1.0) - fx).astype(np.float32) * np.float32(INTER_RESIZE_COEF_SCALE)
    c1 = fx * np.float32(INTER_RESIZE_COEF_SCALE)
    a0 = np.clip(np.rint(c0), -32768, 32767).astype(np.int32)
    a1 = np.clip(np.rint(c1), -32768, 32767).astype(np.int32)
    s1 = np.minimum(s + 1, n_src - 1).astype(np.int32)
    return s, s1, a0, a1


def cv2_resize_linear_u8_np(img: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_LINEAR) for uint8 [H,W] or [H,W,C]
    (infer_two_stage_burr.py:124).  Horizontal pass in int32 (src * alpha, scale 2^11), vertical pass
    ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2 — VResizeLinear<uchar,int,short,...>."""
    dw, dh = int(dsize[0]), int(dsize[1])
    x = np.asarray(img)
    assert x.dtype == np.uint8
    sq = x.ndim == 2
    if sq:
        x = x[:, :, None]
    sh, sw = x.shape[:2]
    xs0, xs1, xa0, xa1 = cv2_linear_tables(sw, dw)
    ys0, ys1, yb0, yb1 = cv2_linear_tables(sh, dh)
    xi = x.astype(np.int32)
    hrow = xi[:, xs0, :] * xa0[None, :, None] + xi[:, xs1, :] * xa1[None, :, None]        # [sh, dw, C]
    S0 = hrow[ys0] >> 4
    S1 = hrow[ys1] >> 4
    out = (((yb0[:, None, None] * S0) >> 16) + ((yb1[:, None, None] * S1) >> 16) + 2) >> 2
    out = np.clip(out, 0, 255).astype(np.uint8)
    return out[:, :, 0] if sq else out


def cv2_nearest_table(n_src: int, n_dst: int):
    """resizeNN: src = min(floor(d * (1 / (n_dst / n_src))), n_src - 1), all in double."""
    inv = np.float64(n_dst) / np.float64(n_src)
    ifx = np.float64(1.0) / inv
    return np.minimum(np.floor(np.arange(n_dst, dtype=np.float64) * ifx).astype(np.int64), n_src - 1).astype(np.int32)


def cv2_resize_nearest_np(img: np.ndarray, dsize) -> np.ndarray:
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_NEAREST) (infer_two_stage_burr.py:307-308)."""
    dw, dh = int(dsize[0]), int(dsize[1])
    x = np.asarray(img)
    return x[cv2_nearest_table(x.shape[0], dh)][:, cv2_nearest_table(x.shape[1], dw)]


FIXED_ROI_512 = (140, 0, 270, 512)      # infer_two_stage_burr.py:29-34 (x1, y1, x2, y2)


def map_roi_to_original_np(original_size, target_size=(512, 512), roi=FIXED_ROI_512):
    """infer_two_stage_burr.py:37-47: scale the fixed 512x512 ROI to the frame size, truncating with int()."""
    ow, oh = original_size
    tw, th = target_size
    sx, sy = ow / tw, oh / th
    return (int(roi[0] * sx), int(roi[1] * sy), int(roi[2] * sx), int(roi[3] * sy))


def clip_to_roi_np(mask_full: np.ndarray, roi) -> np.ndarray:
    """infer_two_stage_burr.py:311-314: zeros outside [y1:y2, x1:x2] (Python slice semantics)."""
    x1, y1, x2, y2 = roi
    out = np.zeros_like(mask_full)
    out[..., y1:y2, x1:x2] = mask_full[..., y1:y2, x1:x2]
    return out


def preprocess_image_np(frame_bgr_u8: np.ndarray, target_size=(512, 512)) -> np.ndarray:
    """infer_two_stage_burr.py:122-127: BGR->RGB, resize, /255, HWC->CHW (float32 [3,H,W])."""
    rgb = frame_bgr_u8[..., ::-1]
    resized = cv2_resize_linear_u8_np(rgb, target_size)
    return np.ascontiguousarray(np.transpose(resized.astype(np.float32) / np.float32(255.0), (2, 0, 1)))


def postprocess_masks_np(pred: np.ndarray, frame_size, roi=None):
    """infer_two_stage_burr.py:303-314 for one [H,W] class-index mask: class-equality masks, nearest resize to
    (width, height), ROI clip."""
    outs = []
    for cls in (1, 2):
        m = (pred == cls).astype(np.uint8)
        full = cv2_resize_nearest_np(m, frame_size)
        outs.append(clip_to_roi_np(full, roi) if roi is not None else full)
    return outs[0], outs[1]


# ----------------------------------------------------------------------------- torch CPU restatement
def torch_forward(sd: dict, x, return_intermediates: bool = False):
    """Same graph through torch.nn.functional on the CPU (fp32) — the ops the reference's
    `--device cpu` path dispatches to.  sd values may be numpy arrays or torch tensors."""
    import torch
    import torch.nn.functional as F

    def T(a):
        return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))

    x = T(x).float()
    if x.shape[2] % 16 or x.shape[3] % 16:
        raise RuntimeError("Sizes of tensors must match: H and W must be multiples of 16")

    def block(x, name):
        for j in (1, 2):
            x = F.conv2d(x, T(sd[f"{name}.conv{j}.weight"]), T(sd[f"{name}.conv{j}.bias"]), padding=1)
            x = F.batch_norm(x, T(sd[f"{name}.bn{j}.running_mean"]), T(sd[f"{name}.bn{j}.running_var"]),
                             T(sd[f"{name}.bn{j}.weight"]), T(sd[f"{name}.bn{j}.bias"]), False, 0.1, BN_EPS)
            x = F.relu(x)
        return x

    def up(x):
        return F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)

    def pool(x):
        return F.max_pool2d(x, 2, 2)

    with torch.no_grad():
        t = {}
        t["x0_0"] = block(x, "conv0_0")
        t["x1_0"] = block(pool(t["x0_0"]), "conv1_0")
        t["x2_0"] = block(pool(t["x1_0"]), "conv2_0")
        t["x3_0"] = block(pool(t["x2_0"]), "conv3_0")
        t["x4_0"] = block(pool(t["x3_0"]), "conv4_0")
        t["x3_1"] = block(torch.cat([t["x3_0"], up(t["x4_0"])], 1), "conv3_1")
        t["x2_2"] = block(torch.cat([t["x2_0"], up(t["x3_1"])], 1), "conv2_2")
        t["x1_3"] = block(torch.cat([t["x1_0"], up(t["x2_2"])], 1), "conv1_3")
        t["x0_4"] = block(torch.cat([t["x0_0"], up(t["x1_3"])], 1), "conv0_4")
        logits = F.conv2d(t["x0_4"], T(sd["final.weight"]), T(sd["final.bias"]))
    if return_intermediates:
        t["logits"] = logits
        return logits.numpy(), {k: v.numpy() for k, v in t.items()}
    return logits.numpy()


def torch_segment(sd: dict, x):
    """Frame-loop tail (infer_two_stage_burr.py:294-300) on the CPU: logits -> softmax -> argmax -> uint8."""
    import torch
    logits = torch.from_numpy(torch_forward(sd, x))
    probs = torch.softmax(logits, dim=1).numpy()
    return np.argmax(probs, axis=1).astype(np.uint8)
