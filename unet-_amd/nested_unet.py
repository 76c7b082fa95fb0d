"""Drop-in for the reference's ``NestedUNet`` (src/models/unetpp.py:28-135) in inference use.

Same constructor keywords, the nn.Module surface the frame loops touch (``.to``, ``.eval``,
``.load_state_dict``, ``__call__``) and the same results; the work is done by hand-written HIP
kernels behind the C ABI of include/unetpp.h.  PyTorch tensors are only I/O buffers (``data_ptr()``)
and the source of the current HIP stream — no torch op runs on the hot path.

    model = NestedUNet(num_classes=3, deep_supervision=True, pretrained_encoder=False).to(device)   # infer_two_stage_burr.py:214
    model.load_state_dict(checkpoint['model'], strict=True); model.eval()                           # :215-217
    outputs = model(img_tensor)                                                                      # :294-297
    pred = model.segment(img_tensor)       # fused replacement of :294-300 (uint8 class-index mask, on device)
"""
from __future__ import annotations

import ctypes
from typing import Optional

import numpy as np

from . import _lib, packing


class NestedUNet:
    _ARCH = _lib.ARCH_NESTED
    _SIZE_MULTIPLE = 16
    def __init__(self, num_classes: int, input_channels: int = 3, deep_supervision: bool = True,
                 pretrained_encoder: bool = False, *, precision: str = "exact", max_batch: int = 16,
                 max_hw=(512, 512), micro_batch: int = 0, streams: int = 1, check_range: bool = False) -> None:
        if pretrained_encoder:
            # unetpp.py:52-65 swaps in a torchvision ResNet50 and downloads ImageNet weights; no
            # north-star caller uses it (infer_two_stage_burr.py:214) and there is no network here.
            raise NotImplementedError("pretrained_encoder=True is unsupported by the MI355X engine")
        if input_channels != 3:
            raise NotImplementedError("input_channels must be 3")
        if precision not in _lib.PRECISIONS:
            raise ValueError("precision must be 'exact', 'exact8' or 'fast'")
        self.num_classes = int(num_classes)
        self.input_channels = int(input_channels)
        self.deep_supervision = bool(deep_supervision)
        self.training = False
        self.precision = precision
        self._max_batch = int(max_batch)
        self._max_hw = (int(max_hw[0]), int(max_hw[1]))
        self._micro_batch = int(micro_batch)
        self._streams = int(streams)
        self._keep_all = False
        self._check_range = bool(check_range)      # debug aid: synchronise and raise after a forward that left the fp16 range
        self._device_index: Optional[int] = None
        self._handle = None
        self._blob: Optional[np.ndarray] = None      # canonical weights (host copy, re-uploaded if the engine is rebuilt)
        self._state_dict = None

    # ------------------------------------------------------------------ nn.Module surface
    def to(self, device):
        import torch
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError(f"device '{dev}': the MI355X engine runs on HIP devices only (no CPU fallback); "
                               "use the reference model for --device cpu")
        idx = dev.index if dev.index is not None else torch.cuda.current_device()
        if self._device_index is not None and idx != self._device_index:
            self._destroy()
        self._device_index = idx
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else f"cuda:{device}")

    def eval(self):
        self.training = False
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("inference-only engine: train() is unsupported")
        return self.eval()

    def load_state_dict(self, state_dict, strict: bool = True):
        state_dict = packing.unwrap_checkpoint(state_dict)
        missing, unexpected = packing.check_state_dict(state_dict, self.num_classes, self.input_channels,
                                                       self.deep_supervision, strict)
        if missing:
            raise RuntimeError("Missing key(s) in state_dict: " + ", ".join(missing))
        self._blob = packing.build_blob(state_dict, self.num_classes, self.input_channels)
        self._state_dict = {k: packing._np(v).copy() for k, v in state_dict.items()}
        if self._handle is not None:
            self._upload()
        return missing, unexpected

    def state_dict(self):
        if self._state_dict is None:
            raise RuntimeError("no weights loaded")
        return dict(self._state_dict)

    def __call__(self, x):
        return self.forward(x)

    # ------------------------------------------------------------------ engine management
    def _err(self, rc: int) -> str:
        lib = _lib.load()
        msg = lib.unetpp_last_error(self._handle)
        return f"unetpp error {rc}: {msg.decode() if msg else ''}"

    def _destroy(self):
        if self._handle is not None:
            _lib.load().unetpp_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self._destroy()
        except Exception:
            pass

    def _ensure_engine(self, b: int, h: int, w: int):
        if self._device_index is None:
            raise RuntimeError("call .to('cuda:N') first: the MI355X engine has no CPU path")
        grow = (self._handle is None or b > self._max_batch or h > self._max_hw[0] or w > self._max_hw[1])
        if not grow:
            return
        self._destroy()
        self._max_batch = max(self._max_batch, b)
        self._max_hw = (max(self._max_hw[0], h), max(self._max_hw[1], w))
        lib = _lib.load()
        cfg = _lib.Config(self.num_classes, self.input_channels, self._max_batch, self._max_hw[0], self._max_hw[1],
                          _lib.PRECISIONS[self.precision], self._device_index,
                          self._micro_batch, self._streams, self._ARCH)
        handle = ctypes.c_void_p()
        rc = lib.unetpp_create(ctypes.byref(cfg), ctypes.byref(handle))
        if rc != 0:
            msg = lib.unetpp_last_error(None)
            raise RuntimeError(f"unetpp_create failed ({rc}): {msg.decode() if msg else ''}")
        self._handle = handle
        if self._keep_all:
            lib.unetpp_debug_keep_intermediates(self._handle, 1)
        if self._blob is not None:
            self._upload()

    def _upload(self):
        lib = _lib.load()
        rc = lib.unetpp_load_weights(self._handle, self._blob.ctypes.data_as(ctypes.c_void_p), self._blob.nbytes)
        if rc != 0:
            raise RuntimeError(self._err(rc))

    def _check_and_build_blob(self, state_dict):
        """strict key check + canonical blob of this architecture (rank 0 of sharding.load_replicated)."""
        state_dict = packing.unwrap_checkpoint(state_dict)
        packing.check_state_dict(state_dict, self.num_classes, self.input_channels, self.deep_supervision, strict=True)
        return packing.build_blob(state_dict, self.num_classes, self.input_channels)

    def load_weights_from_device_blob(self, blob_tensor):
        """After an RCCL broadcast: `blob_tensor` is a uint8 CUDA tensor holding the canonical blob."""
        import torch
        self._ensure_engine(1, self._SIZE_MULTIPLE, self._SIZE_MULTIPLE)
        rc = _lib.load().unetpp_load_weights_device(self._handle, ctypes.c_void_p(blob_tensor.data_ptr()),
                                                    blob_tensor.numel(), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        if rc != 0:
            raise RuntimeError(self._err(rc))
        self._blob = blob_tensor.cpu().numpy()

    # ------------------------------------------------------------------ the hot path
    def _run(self, x, want_logits: bool, want_mask: bool, want_class_masks: bool, want_probs: bool = False,
             rule: str = "argmax", params=(0.0, 0.0, 0.0, 0.0)):
        import torch
        if self.training:
            raise RuntimeError("engine is inference-only")
        if not isinstance(x, torch.Tensor) or not x.is_cuda:
            raise RuntimeError("input must be a CUDA (HIP) tensor on the engine's device")
        if self._blob is None:
            raise RuntimeError("load_state_dict() must be called before forward")
        if x.dtype == torch.float32:
            if x.dim() != 4 or x.shape[1] != self.input_channels:
                raise RuntimeError(f"expected input [B,{self.input_channels},H,W], got {tuple(x.shape)}")
            b, _, h, w = x.shape
            fmt = _lib.IN_F32_NCHW
        elif x.dtype == torch.uint8:
            if x.dim() != 4 or x.shape[3] != 3:
                raise RuntimeError(f"expected uint8 frames [B,H,W,3] (BGR), got {tuple(x.shape)}")
            b, h, w, _ = x.shape
            fmt = _lib.IN_U8_NHWC_BGR
        else:
            raise RuntimeError(f"unsupported input dtype {x.dtype}")
        if h % self._SIZE_MULTIPLE or w % self._SIZE_MULTIPLE:
            # same failure the reference hits inside torch.cat (unetpp.py:112-116) for such sizes
            raise RuntimeError(f"Sizes of tensors must match: H={h}, W={w} must be multiples of {self._SIZE_MULTIPLE}")
        if self._device_index is None:
            self.to(x.device)
        if x.device.index != self._device_index:
            raise RuntimeError(f"input on {x.device}, engine on cuda:{self._device_index}")
        x = x.contiguous()
        self._ensure_engine(b, h, w)
        dev = x.device
        logits = torch.empty((b, self.num_classes, h, w), dtype=torch.float32, device=dev) if want_logits else None
        mask = torch.empty((b, h, w), dtype=torch.uint8, device=dev) if want_mask else None
        cable = torch.empty((b, h, w), dtype=torch.uint8, device=dev) if want_class_masks else None
        tape = torch.empty((b, h, w), dtype=torch.uint8, device=dev) if want_class_masks else None
        probs = torch.empty((b, self.num_classes, h, w), dtype=torch.float32, device=dev) if want_probs else None
        p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
        stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        if rule not in _lib.RULES:
            raise ValueError(f"rule must be one of {sorted(_lib.RULES)}")
        outs = _lib.Outputs(p(logits), p(probs), p(mask), p(cable), p(tape), _lib.RULES[rule], *[float(v) for v in params])
        rc = _lib.load().unetpp_forward_ex(self._handle, p(x), fmt, b, h, w, ctypes.byref(outs), stream)
        if rc != 0:
            raise RuntimeError(self._err(rc))
        if self._check_range:
            self.raise_on_range_error()
        if want_probs:
            return logits, mask, cable, tape, probs
        return logits, mask, cable, tape

    # ------------------------------------------------------------------ value-range status
    def status(self, clear: bool = False) -> int:
        """Sticky range flags of the engine (include/unetpp.h: UNETPP_STATUS_OVERFLOW = 1, UNETPP_STATUS_NAN = 2):
        set when an activation or input value did not fit the fp16 planes the engine stores activations in — the
        fp32 reference (unetpp.py:23-26) has no such limit, so a non-zero status means the results may differ from
        it.  Synchronises the device."""
        if self._handle is None:
            return 0
        flags = ctypes.c_uint32(0)
        rc = _lib.load().unetpp_status(self._handle, ctypes.byref(flags), 1 if clear else 0)
        if rc != 0:
            raise RuntimeError(self._err(rc))
        return int(flags.value)

    def raise_on_range_error(self):
        """RuntimeError if any forward since the last check left the fp16 range (clears the flags)."""
        flags = self.status(clear=True)
        if flags:
            what = [n for bit, n in ((_lib.STATUS_OVERFLOW, "activation/input beyond +-65504 clamped"),
                                     (_lib.STATUS_NAN, "NaN encountered")) if flags & bit]
            raise RuntimeError("unetpp: value range of the fp16 activation planes exceeded (" + "; ".join(what) +
                               "): results differ from the fp32 reference")

    def forward(self, x):
        """NestedUNet.forward in eval mode (unetpp.py:93-135): float32 [B,3,H,W] -> float32 logits [B,C,H,W]."""
        return self._run(x, True, False, False)[0]

    def segment(self, x, return_logits: bool = False, return_class_masks: bool = False):
        """Fused model call + softmax/argmax/uint8 (+ class masks) of infer_two_stage_burr.py:294-304.
        x: float32 [B,3,H,W] in [0,1] or uint8 [B,H,W,3] BGR frames at model resolution."""
        logits, mask, cable, tape = self._run(x, return_logits, True, return_class_masks)
        out = (mask,)
        if return_class_masks:
            out += (cable, tape)
        if return_logits:
            out += (logits,)
        return out[0] if len(out) == 1 else out

    def segment_thresholded(self, x, rule: str = "thresholded_argmax", t_cable: float = 0.45, t_tape: float = 0.50,
                            bg_margin: float = 0.15, ct_margin: float = 0.10, return_probs: bool = False):
        """The thresholded frame loops' tail on the device: probs = softmax_np(outputs) then
        `thresholded_argmax` (infer_video_3class_best.py:56-83, infer_video_strict.py:36-63),
        `strict_bg_check` (infer_video_fixed.py:35-83: bg_margin is the background-probability ceiling) or
        `exclusive` (infer_video_robust.py:70-99).  Returns (mask_cable, mask_tape[, probs[B,C,H,W]]) on device."""
        r = self._run(x, False, False, True, return_probs, rule, (t_cable, t_tape, bg_margin, ct_margin))
        return (r[2], r[3], r[4]) if return_probs else (r[2], r[3])

    def mask_stats(self, mask):
        """Device-side reductions of a uint8 class-index mask [B,H,W] (e.g. from segment()):
        returns (counts int64 [B,C], widths float32 [B,C,H]) where counts[b,c] = np.sum(mask[b]==c)
        (infer_two_stage_burr.py:333-334) and widths[b,c,y] = xs.max()-xs.min()+1 over the columns of class c
        in row y, 0 for empty rows (_compute_width_per_row, src/utils/geometry_enhanced.py:45-74, before its
        optional smoothing).  Only B*C*(2H+1) integers cross to the caller instead of the mask."""
        import torch
        if not (isinstance(mask, torch.Tensor) and mask.is_cuda and mask.dtype == torch.uint8 and mask.dim() == 3):
            raise RuntimeError("mask must be a uint8 CUDA tensor [B,H,W]")
        if self._handle is None:
            raise RuntimeError("engine not initialised: run a forward first")
        mask = mask.contiguous()
        b, h, w = mask.shape
        c = self.num_classes
        counts = torch.empty((b, c), dtype=torch.int32, device=mask.device)
        rmin = torch.empty((b, c, h), dtype=torch.int32, device=mask.device)
        rmax = torch.empty((b, c, h), dtype=torch.int32, device=mask.device)
        p = lambda t: ctypes.c_void_p(t.data_ptr())
        rc = _lib.load().unetpp_mask_stats(self._handle, p(mask), b, h, w, p(counts), p(rmin), p(rmax),
                                           ctypes.c_void_p(torch.cuda.current_stream(mask.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(self._err(rc))
        widths = torch.where(rmax >= 0, (rmax - rmin + 1), torch.zeros_like(rmax)).to(torch.float32)
        return counts.to(torch.int64), widths

    def resize_frames(self, frames, size_hw):
        """cv2.resize(frame, (W, H), interpolation=cv2.INTER_LINEAR) for uint8 CUDA frames [B,h,w,C] -> [B,H,W,C]
        (preprocess_image, infer_two_stage_burr.py:124).  Chain with segment(): the BGR->RGB swap and /255 run
        inside the engine's first kernel."""
        import torch
        if not (isinstance(frames, torch.Tensor) and frames.is_cuda and frames.dtype == torch.uint8 and frames.dim() == 4):
            raise RuntimeError("frames must be a uint8 CUDA tensor [B,H,W,C]")
        H, W = int(size_hw[0]), int(size_hw[1])
        self._ensure_engine(1, self._SIZE_MULTIPLE, self._SIZE_MULTIPLE)
        frames = frames.contiguous()
        b, h, w, c = frames.shape
        out = torch.empty((b, H, W, c), dtype=torch.uint8, device=frames.device)
        rc = _lib.load().unetpp_resize_linear_u8(self._handle, ctypes.c_void_p(frames.data_ptr()), b, h, w, c,
                                                 ctypes.c_void_p(out.data_ptr()), H, W,
                                                 ctypes.c_void_p(torch.cuda.current_stream(frames.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(self._err(rc))
        return out

    def resize_masks(self, pred, frame_size_wh, match_class: int = -1, roi=None):
        """infer_two_stage_burr.py:303-314 on the device for a uint8 CUDA mask [B,H,W]: optional
        `(pred == match_class)`, cv2.resize(..., (width, height), INTER_NEAREST), zeros outside
        roi = (x1, y1, x2, y2).  Returns uint8 [B,height,width]."""
        import torch
        if not (isinstance(pred, torch.Tensor) and pred.is_cuda and pred.dtype == torch.uint8 and pred.dim() == 3):
            raise RuntimeError("pred must be a uint8 CUDA tensor [B,H,W]")
        fw, fh = int(frame_size_wh[0]), int(frame_size_wh[1])
        x1, y1, x2, y2 = (0, 0, fw, fh) if roi is None else (int(v) for v in roi)
        self._ensure_engine(1, self._SIZE_MULTIPLE, self._SIZE_MULTIPLE)
        pred = pred.contiguous()
        b, h, w = pred.shape
        out = torch.empty((b, fh, fw), dtype=torch.uint8, device=pred.device)
        rc = _lib.load().unetpp_resize_nearest_roi_u8(self._handle, ctypes.c_void_p(pred.data_ptr()), b, h, w,
                                                      int(match_class), ctypes.c_void_p(out.data_ptr()), fh, fw,
                                                      x1, y1, x2, y2,
                                                      ctypes.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream))
        if rc != 0:
            raise RuntimeError(self._err(rc))
        return out

    def predict_proba(self, x):
        """softmax(model(x), dim=1) as float32 [B,C,H,W] on the device (one fused pass)."""
        return self._run(x, False, False, False, True)[4]

    # ------------------------------------------------------------------ measurement / debug hooks
    def workspace_bytes(self) -> int:
        return int(_lib.load().unetpp_workspace_bytes(self._handle)) if self._handle else 0

    def profile(self, on: bool = True):
        _lib.load().unetpp_profile_enable(self._handle, 1 if on else 0)

    def profile_read(self):
        """[(launch name, ms, algorithmic flops, min HBM bytes)] of the last forward (profiling on)."""
        lib = _lib.load()
        n = lib.unetpp_profile_count(self._handle)
        ms = (ctypes.c_float * max(n, 1))()
        got = lib.unetpp_profile_read(self._handle, ms, n)
        if got < 0:
            raise RuntimeError(self._err(got))
        out = []
        for i in range(got):
            fl, by = ctypes.c_double(), ctypes.c_double()
            lib.unetpp_profile_work(self._handle, i, ctypes.byref(fl), ctypes.byref(by))
            out.append((lib.unetpp_profile_name(self._handle, i).decode(), float(ms[i]), fl.value, by.value))
        return out

    def debug_keep_intermediates(self, on: bool = True):
        """Materialise x0_4 and run the head unfused (needed before debug_activation('x0_4'))."""
        self._keep_all = bool(on)                    # also applied to an engine that is (re)built later
        if self._handle is not None:
            _lib.load().unetpp_debug_keep_intermediates(self._handle, 1 if on else 0)

    def _node_shape(self, name: str):
        lvl = int(name[1])
        return (32, 64, 128, 256, 512)[lvl], lvl

    def debug_activation(self, name: str, b: int, h: int, w: int) -> np.ndarray:
        """float32 [b,C,h',w'] copy of an intermediate node ('x0_0'..'x4_0','x3_1','x2_2','x1_3','x0_4')."""
        c, lvl = self._node_shape(name)
        out = np.empty((b, c, h >> lvl, w >> lvl), dtype=np.float32)
        n = _lib.load().unetpp_debug_read(self._handle, name.encode(), out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), out.size)
        if n < 0:
            raise RuntimeError(self._err(int(n)))
        if n != out.size:
            raise RuntimeError(f"debug_read returned {n} floats, expected {out.size}")
        return out


class SimpleUNet(NestedUNet):
    """Drop-in for the reference's ``SimpleUNet`` (src/models/simple_unet.py:20-128), the
    plain 4-level U-Net of infer_video_simple.py:67: same constructor keywords (num_classes=7, num_channels=3),
    same state_dict keys (enc1.0 ... final), logits from ``model(x)``; ``predict_proba`` replaces the
    ``torch.softmax(output, dim=1)`` of infer_video_simple.py:96.  H and W must be multiples of 8."""
    _ARCH = _lib.ARCH_SIMPLE
    _SIZE_MULTIPLE = 8

    def __init__(self, num_classes: int = 7, num_channels: int = 3, *, precision: str = "exact", max_batch: int = 16,
                 max_hw=(256, 256), micro_batch: int = 0, streams: int = 1, check_range: bool = False) -> None:
        super().__init__(num_classes, num_channels, False, False, precision=precision, max_batch=max_batch,
                         max_hw=max_hw, micro_batch=micro_batch, streams=streams, check_range=check_range)
        self.num_channels = int(num_channels)

    def load_state_dict(self, state_dict, strict: bool = True):
        state_dict = packing.unwrap_checkpoint(state_dict)
        missing, unexpected = packing.check_simple_state_dict(state_dict, self.num_classes, self.num_channels, strict)
        if missing:
            raise RuntimeError("Missing key(s) in state_dict: " + ", ".join(missing))
        self._blob = packing.build_simple_blob(state_dict, self.num_classes, self.num_channels)
        self._state_dict = {k: packing._np(v).copy() for k, v in state_dict.items()}
        if self._handle is not None:
            self._upload()
        return missing, unexpected

    def _check_and_build_blob(self, state_dict):
        state_dict = packing.unwrap_checkpoint(state_dict)
        packing.check_simple_state_dict(state_dict, self.num_classes, self.num_channels, strict=True)
        return packing.build_simple_blob(state_dict, self.num_classes, self.num_channels)

    def _node_shape(self, name: str):
        lvl = int(name[3]) - 1                      # 'enc1'..'enc4', 'dec1'..'dec3'
        return (64, 128, 256, 512)[lvl], lvl
