"""Host side of load_state_dict: strict key check, BatchNorm folding (float64) and the canonical
weight blob that unetpp_load_weights() consumes (layout: include/unetpp.h).

Folding:  y = relu(bn(conv(x)))  with eval-mode BN (reference src/models/unetpp.py:17-26)
    s = gamma / sqrt(running_var + eps);  W' = W * s[:,None,None,None];  b' = (b - running_mean) * s + beta
"""
from __future__ import annotations

import numpy as np

from .manifest import conv_blocks, simple_unet_manifest, state_dict_manifest

BN_EPS = 1e-5
BLOB_MAGIC = 0x50504E55  # 'UNPP'
BLOB_VERSION = 1


def _np(v):
    if hasattr(v, "detach"):
        v = v.detach().cpu().numpy()
    return np.asarray(v)


def infer_num_classes(state_dict) -> int:
    """infer_video_refactored.py:59-89 reads it off final.weight."""
    return int(_np(state_dict["final.weight"]).shape[0])


def unwrap_checkpoint(ckpt):
    """checkpoint['model'] (infer_two_stage_burr.py:216) / ['model_state_dict'] / bare dict (infer_video.py:148-153)."""
    if isinstance(ckpt, dict):
        if "model" in ckpt and isinstance(ckpt["model"], dict):
            return ckpt["model"]
        if "model_state_dict" in ckpt and isinstance(ckpt["model_state_dict"], dict):
            return ckpt["model_state_dict"]
    return ckpt


def check_state_dict(state_dict, num_classes: int, in_channels: int, deep_supervision: bool, strict: bool = True):
    """Mirror of nn.Module.load_state_dict(strict=...): returns (missing, unexpected); raises RuntimeError
    with the same wording when strict and anything is off, or on a shape mismatch."""
    manifest = state_dict_manifest(num_classes, in_channels, deep_supervision)
    expected = {k: tuple(s) for k, s, _ in manifest}
    missing = [k for k in expected if k not in state_dict]
    unexpected = [k for k in state_dict if k not in expected]
    errs = []
    for k, shp in expected.items():
        if k in state_dict and tuple(_np(state_dict[k]).shape) != shp:
            errs.append(f"size mismatch for {k}: copying a param with shape {tuple(_np(state_dict[k]).shape)} "
                        f"from checkpoint, the shape in current model is {shp}.")
    if strict and (missing or unexpected):
        if unexpected:
            errs.insert(0, "Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
        if missing:
            errs.insert(0, "Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
    if errs:
        raise RuntimeError("Error(s) in loading state_dict for NestedUNet:\n\t" + "\n\t".join(errs))
    return missing, unexpected


def fold_conv_bn(w, b, gamma, beta, mean, var, eps: float = BN_EPS):
    w = _np(w).astype(np.float64); b = _np(b).astype(np.float64)
    s = _np(gamma).astype(np.float64) / np.sqrt(_np(var).astype(np.float64) + eps)
    wf = w * s[:, None, None, None]
    bf = (b - _np(mean).astype(np.float64)) * s + _np(beta).astype(np.float64)
    return wf.astype(np.float32), bf.astype(np.float32)


def folded_layers(state_dict, in_channels: int = 3):
    """[(name, W'[Co,Ci,3,3] f32, b'[Co] f32)] for the 18 3x3 convs in forward order, then the 1x1 head."""
    out = []
    for name, _, _ in conv_blocks(in_channels):
        for j in (1, 2):
            wf, bf = fold_conv_bn(state_dict[f"{name}.conv{j}.weight"], state_dict[f"{name}.conv{j}.bias"],
                                  state_dict[f"{name}.bn{j}.weight"], state_dict[f"{name}.bn{j}.bias"],
                                  state_dict[f"{name}.bn{j}.running_mean"], state_dict[f"{name}.bn{j}.running_var"])
            out.append((f"{name}.conv{j}", wf, bf))
    out.append(("final", _np(state_dict["final.weight"]).astype(np.float32), _np(state_dict["final.bias"]).astype(np.float32)))
    return out


def build_blob(state_dict, num_classes: int, in_channels: int = 3) -> np.ndarray:
    """Canonical blob (uint8 array): 32-byte header + fp32 payload; ds* heads and num_batches_tracked are dropped
    (deep-supervision heads run only in train mode, unetpp.py:121-133)."""
    layers = folded_layers(state_dict, in_channels)
    header = np.array([BLOB_MAGIC, BLOB_VERSION, num_classes, in_channels, len(layers), 0, 0, 0], dtype=np.uint32)   # arch 0
    parts = [header.view(np.uint8)]
    for _, w, b in layers:
        parts.append(np.ascontiguousarray(w, dtype=np.float32).ravel().view(np.uint8))
        parts.append(np.ascontiguousarray(b, dtype=np.float32).ravel().view(np.uint8))
    return np.concatenate(parts)


# ---------------------------------------------------------------------------- SimpleUNet
def check_simple_state_dict(state_dict, num_classes: int, num_channels: int = 3, strict: bool = True):
    """nn.Module.load_state_dict(strict=...) semantics for SimpleUNet (src/models/simple_unet.py:30-92)."""
    expected = {k: tuple(s) for k, s, _ in simple_unet_manifest(num_classes, num_channels)}
    missing = [k for k in expected if k not in state_dict]
    unexpected = [k for k in state_dict if k not in expected]
    errs = []
    for k, shp in expected.items():
        if k in state_dict and tuple(_np(state_dict[k]).shape) != shp:
            errs.append(f"size mismatch for {k}: copying a param with shape {tuple(_np(state_dict[k]).shape)} "
                        f"from checkpoint, the shape in current model is {shp}.")
    if strict and (missing or unexpected):
        if unexpected:
            errs.insert(0, "Unexpected key(s) in state_dict: " + ", ".join(f'"{k}"' for k in unexpected) + ". ")
        if missing:
            errs.insert(0, "Missing key(s) in state_dict: " + ", ".join(f'"{k}"' for k in missing) + ". ")
    if errs:
        raise RuntimeError("Error(s) in loading state_dict for SimpleUNet:\n\t" + "\n\t".join(errs))
    return missing, unexpected


def build_simple_blob(state_dict, num_classes: int, num_channels: int = 3) -> np.ndarray:
    """Canonical blob for arch 1: the state_dict's tensors in definition order (weights then bias per layer),
    fp32, nothing folded (SimpleUNet has no BatchNorm)."""
    man = simple_unet_manifest(num_classes, num_channels)
    n_layers = len(man) // 2
    header = np.array([BLOB_MAGIC, BLOB_VERSION, num_classes, num_channels, n_layers, 1, 0, 0], dtype=np.uint32)
    parts = [header.view(np.uint8)]
    for key, _, _ in man:
        parts.append(np.ascontiguousarray(_np(state_dict[key]), dtype=np.float32).ravel().view(np.uint8))
    return np.concatenate(parts)
