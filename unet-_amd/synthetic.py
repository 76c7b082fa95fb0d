"""Deterministic synthetic weights and frames (NumPy only, PCG64).

The reference ships no checkpoint and no sample video (its .gitignore:30-40 excludes them), so
every parity test and every bench line runs on synthetic data generated here.  The generator is
NumPy-only so that the GPU box (where /root/reference does not exist) regenerates bit-identical
weights and frames from the seed.

State-dict layout mirrors ``NestedUNet.state_dict()`` of the reference
(src/models/unetpp.py:13-26 ConvBlock, :66-91 NestedUNet members): per ConvBlock
``{conv1,conv2}.{weight[Co,Ci,3,3],bias[Co]}``, ``{bn1,bn2}.{weight,bias,running_mean,
running_var}[Co]`` + ``num_batches_tracked`` (int64 scalar); ``final.{weight[C,32,1,1],bias[C]}``;
with deep supervision also ``ds3_1/ds2_2/ds1_3.{weight,bias}``.

Default PyTorch init gives a degenerate net (every pixel argmaxes to one class), so weights are
He-normal with perturbed BN statistics: all classes appear and logits span roughly [-3, 2].
"""
from __future__ import annotations

import numpy as np

from .manifest import NB_FILTER, SIMPLE_WIDTHS, conv_blocks, simple_unet_manifest, state_dict_manifest  # noqa: F401 (re-exported)


def make_state_dict(num_classes: int = 3, in_channels: int = 3, deep_supervision: bool = True,
                    seed: int = 0) -> dict:
    """He-normal conv weights, N(0,0.05^2) biases, BN gamma/var ~ U[0.75,1.25], beta/mean ~ N(0,0.2^2)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = {}
    for key, shape, dtype in state_dict_manifest(num_classes, in_channels, deep_supervision):
        leaf = key.split(".")[-1]
        parent = key.split(".")[-2]
        if dtype == "int64":
            sd[key] = np.array(100, dtype=np.int64)
        elif leaf == "weight" and len(shape) == 4:
            fan_in = shape[1] * shape[2] * shape[3]
            sd[key] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif leaf == "bias" and not parent.startswith("bn"):
            sd[key] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
        elif leaf in ("weight", "running_var"):          # BN gamma, running variance
            sd[key] = rng.uniform(0.75, 1.25, shape).astype(np.float32)
        else:                                            # BN beta, running mean
            sd[key] = (rng.standard_normal(shape) * 0.2).astype(np.float32)
    return sd


def make_simple_state_dict(num_classes: int = 7, num_channels: int = 3, seed: int = 0) -> dict:
    """He-normal weights (fan_in of the forward contraction), N(0, 0.05^2) biases."""
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    sd = {}
    for key, shape, _ in simple_unet_manifest(num_classes, num_channels):
        if key.endswith("weight"):
            if key.startswith("up"):
                fan_in = shape[0]                      # each output pixel sees one tap of every input channel
            else:
                fan_in = shape[1] * shape[2] * shape[3]
            sd[key] = (rng.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        else:
            sd[key] = (rng.standard_normal(shape) * 0.05).astype(np.float32)
    return sd


def _bilinear_up(a: np.ndarray, h: int, w: int) -> np.ndarray:
    """Plain half-pixel bilinear resize of a [h0,w0,c] float array (frame synthesis only)."""
    h0, w0 = a.shape[:2]
    ys = np.clip((np.arange(h) + 0.5) * h0 / h - 0.5, 0, h0 - 1)
    xs = np.clip((np.arange(w) + 0.5) * w0 / w - 0.5, 0, w0 - 1)
    y0 = np.floor(ys).astype(int); y1 = np.minimum(y0 + 1, h0 - 1); fy = (ys - y0)[:, None, None]
    x0 = np.floor(xs).astype(int); x1 = np.minimum(x0 + 1, w0 - 1); fx = (xs - x0)[None, :, None]
    top = a[y0][:, x0] * (1 - fx) + a[y0][:, x1] * fx
    bot = a[y1][:, x0] * (1 - fx) + a[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def make_frame_u8(h: int, w: int, index: int = 0, kind: str = "smooth", seed: int = 1234) -> np.ndarray:
    """uint8 HWC 'BGR' frame, already at model resolution (the cv2 resize of
    infer_two_stage_burr.py:124 is outside the engine contract).

    kind='uniform': i.i.d. uniform bytes; kind='smooth': low-frequency field + 20 % noise
    (video-like: large flat regions, so near-tie pixels are rarer but class regions are coherent).
    """
    rng = np.random.Generator(np.random.PCG64(seed + index))
    if kind == "uniform":
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    lo = rng.uniform(0.0, 255.0, (max(h // 32, 2), max(w // 32, 2), 3))
    noise = rng.uniform(0.0, 255.0, (h, w, 3))
    f = 0.8 * _bilinear_up(lo, h, w) + 0.2 * noise
    return np.ascontiguousarray(np.clip(np.rint(f), 0, 255).astype(np.uint8))     # C order: raw memcpy-able


def make_frames_u8(b: int, h: int, w: int, kind: str = "smooth", seed: int = 1234, first: int = 0) -> np.ndarray:
    return np.ascontiguousarray(np.stack([make_frame_u8(h, w, first + i, kind, seed) for i in range(b)]))


def frames_to_chw_f32(frames_u8: np.ndarray) -> np.ndarray:
    """uint8 [B,H,W,3] BGR -> float32 [B,3,H,W] RGB in [0,1]: the resize-free part of
    preprocess_image (infer_two_stage_burr.py:122-127): BGR->RGB, astype(float32)/255.0, HWC->CHW."""
    rgb = frames_u8[..., ::-1]
    x = rgb.astype(np.float32) / np.float32(255.0)
    return np.ascontiguousarray(np.transpose(x, (0, 3, 1, 2)))


def make_trained_like_state_dict(num_classes: int = 3, in_channels: int = 3, deep_supervision: bool = True,
                                 seed: int = 2, big_gain: float = 3.0) -> dict:
    """`make_state_dict` with the BatchNorm statistics a trained checkpoint shows and He-init never does
    (probes BN folding and the fp16 dynamic range, reference unetpp.py:17-26): in every BN a few
      * dead channels   conv weight/bias and running_mean x 1e-4, running_var = 1e-8 (eps dominates the fold),
      * negative gamma, zero gamma, and large gamma (x big_gain) channels."""
    sd = make_state_dict(num_classes, in_channels, deep_supervision, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    for k in list(sd):
        if not k.endswith("running_var"):
            continue
        bn = k[:-len("running_var")]                  # 'conv0_0.bn1.'
        conv = bn.replace(".bn", ".conv")
        n = sd[k].shape[0]
        idx = rng.permutation(n)
        a, b = max(1, n // 16), max(1, n // 32)
        dead, neg, zero, big = idx[:a], idx[a:2 * a], idx[2 * a:2 * a + b], idx[2 * a + b:2 * a + 2 * b]
        sd[k][dead] = 1e-8
        sd[conv + "weight"][dead] *= 1e-4
        sd[conv + "bias"][dead] *= 1e-4
        sd[bn + "running_mean"][dead] *= 1e-4
        sd[bn + "weight"][neg] *= -1
        sd[bn + "weight"][zero] = 0
        sd[bn + "weight"][big] *= big_gain
    return sd
