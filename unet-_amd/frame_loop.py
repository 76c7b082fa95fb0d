"""The model-adjacent lines of the reference frame loop (infer_two_stage_burr.py:122-127, 292-304)
with the cv2 stages left out: everything between "a BGR frame at model resolution" and
"uint8 class masks".  The cv2.resize calls either side stay on the host, outside the engine contract.
"""
from __future__ import annotations

import numpy as np


def preprocess_frames(frames_bgr_u8: np.ndarray) -> np.ndarray:
    """Resize-free part of preprocess_image (infer_two_stage_burr.py:122-127) for a batch of frames:
    uint8 [B,H,W,3] BGR -> float32 [B,3,H,W] RGB in [0,1]."""
    rgb = frames_bgr_u8[..., ::-1]
    return np.ascontiguousarray(np.transpose(rgb.astype(np.float32) / np.float32(255.0), (0, 3, 1, 2)))


def segment_frames(model, frames_bgr_u8, device=None):
    """infer_two_stage_burr.py:292-304 for B frames at once.

    `frames_bgr_u8` is a uint8 [B,H,W,3] BGR array/tensor already at model resolution.  Returns
    (pred, mask_cable, mask_tape) as uint8 CUDA tensors [B,H,W]; the BGR->RGB swap, /255, layout
    change, forward, softmax/argmax and the class-equality masks all run inside one engine call."""
    import torch
    x = frames_bgr_u8
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.to(device if device is not None else f"cuda:{model._device_index or 0}", non_blocking=True)
    pred, cable, tape = model.segment(x, return_class_masks=True)
    return pred, cable, tape
