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

NB_FILTER = (32, 64, 128, 256, 512)  # src/models/unetpp.py:49

# (name, in_channels, out_channels) in the order the forward uses them (unetpp.py:104-116)
def conv_blocks(in_channels: int = 3):
    f = NB_FILTER
    return [
        ("conv0_0", in_channels, f[0]),
        ("conv1_0", f[0], f[1]),
        ("conv2_0", f[1], f[2]),
        ("conv3_0", f[2], f[3]),
        ("conv4_0", f[3], f[4]),
        ("conv3_1", f[3] + f[4], f[3]),
        ("conv2_2", f[2] + f[3], f[2]),
        ("conv1_3", f[1] + f[2], f[1]),
        ("conv0_4", f[0] + f[1], f[0]),
    ]


def state_dict_manifest(num_classes: int, in_channels: int = 3, deep_supervision: bool = True):
    """Ordered (key, shape, dtype) list identical to the reference module's state_dict()."""
    ordered = []
    for name, ci, co in conv_blocks(in_channels):
        for j, cin in ((1, ci), (2, co)):       # module order: conv1, bn1, conv2, bn2
            ordered.append((f"{name}.conv{j}.weight", (co, cin, 3, 3), "float32"))
            ordered.append((f"{name}.conv{j}.bias", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.weight", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.bias", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.running_mean", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.running_var", (co,), "float32"))
            ordered.append((f"{name}.bn{j}.num_batches_tracked", (), "int64"))
    ordered.append(("final.weight", (num_classes, NB_FILTER[0], 1, 1), "float32"))
    ordered.append(("final.bias", (num_classes,), "float32"))
    if deep_supervision:
        for nm, c in (("ds3_1", NB_FILTER[3]), ("ds2_2", NB_FILTER[2]), ("ds1_3", NB_FILTER[1])):
            ordered.append((f"{nm}.weight", (num_classes, c, 1, 1), "float32"))
            ordered.append((f"{nm}.bias", (num_classes,), "float32"))
    return ordered


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


SIMPLE_WIDTHS = (64, 128, 256, 512)  # src/models/simple_unet.py:32-57


def simple_unet_manifest(num_classes: int = 7, num_channels: int = 3):
    """Ordered (key, shape, dtype) list identical to SimpleUNet.state_dict() (src/models/simple_unet.py:30-92):
    enc{1..4}.{0,2}, up3, up2, up1 (ConvTranspose2d: weight [Cin, Cout, 2, 2]), dec{3,2,1}.{0,2}, final."""
    w = SIMPLE_WIDTHS
    out = []
    for l in range(4):
        ci = num_channels if l == 0 else w[l - 1]
        out += [(f"enc{l+1}.0.weight", (w[l], ci, 3, 3), "float32"), (f"enc{l+1}.0.bias", (w[l],), "float32"),
                (f"enc{l+1}.2.weight", (w[l], w[l], 3, 3), "float32"), (f"enc{l+1}.2.bias", (w[l],), "float32")]
    for l in (2, 1, 0):
        out += [(f"up{l+1}.weight", (w[l + 1], w[l], 2, 2), "float32"), (f"up{l+1}.bias", (w[l],), "float32")]
    for l in (2, 1, 0):
        out += [(f"dec{l+1}.0.weight", (w[l], 2 * w[l], 3, 3), "float32"), (f"dec{l+1}.0.bias", (w[l],), "float32"),
                (f"dec{l+1}.2.weight", (w[l], w[l], 3, 3), "float32"), (f"dec{l+1}.2.bias", (w[l],), "float32")]
    out += [("final.weight", (num_classes, w[0], 1, 1), "float32"), ("final.bias", (num_classes,), "float32")]
    return out


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
