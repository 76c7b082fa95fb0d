"""The model-adjacent lines of the reference frame loop (infer_two_stage_burr.py:37-47, 122-127, 292-314):
everything between "a BGR video frame" and "uint8 class masks at frame size, clipped to the ROI".
`segment_frames` starts from frames already at model resolution; `process_frames` also runs the two
cv2.resize steps either side on the device (those two restate OpenCV's published
algorithm and are parity-unpinned, see oracle/unetpp_oracle.py).
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


FIXED_ROI_512 = {"x1": 140, "y1": 0, "x2": 270, "y2": 512}      # infer_two_stage_burr.py:29-34


def map_roi_to_original(original_size, target_size=(512, 512), roi=None):
    """infer_two_stage_burr.py:37-47: the fixed 512x512 ROI scaled to the frame size (width, height)."""
    orig_w, orig_h = original_size
    target_w, target_h = target_size
    scale_x = orig_w / target_w
    scale_y = orig_h / target_h
    roi = FIXED_ROI_512 if roi is None else roi
    return (int(roi["x1"] * scale_x), int(roi["y1"] * scale_y), int(roi["x2"] * scale_x), int(roi["y2"] * scale_y))


def process_frames(model, frames_bgr_u8, target_size=(512, 512), roi="fixed", device=None):
    """infer_two_stage_burr.py:292-314 for B raw frames at once, every step on the device:
    preprocess_image (BGR->RGB, cv2.resize INTER_LINEAR to `target_size` = (width, height), /255, CHW), the
    model, softmax/argmax, `(pred == 1)`, `(pred == 2)`, cv2.resize INTER_NEAREST back to the frame size and the
    ROI clip.  `roi`: "fixed" (map_roi_to_original of the frame size), None, or (x1, y1, x2, y2) in frame pixels.
    Returns (pred uint8 [B,H,W] at model resolution, mask_cable, mask_tape uint8 [B,frame_h,frame_w])."""
    import torch
    x = frames_bgr_u8
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(x))
    if not x.is_cuda:
        x = x.to(device if device is not None else f"cuda:{model._device_index or 0}", non_blocking=True)
    fh, fw = int(x.shape[1]), int(x.shape[2])
    tw, th = int(target_size[0]), int(target_size[1])
    resized = x if (fh, fw) == (th, tw) else model.resize_frames(x, (th, tw))
    pred = model.segment(resized)
    if roi == "fixed":
        roi = map_roi_to_original((fw, fh), (tw, th))
    cable = model.resize_masks(pred, (fw, fh), match_class=1, roi=roi)
    tape = model.resize_masks(pred, (fw, fh), match_class=2, roi=roi)
    return pred, cable, tape
