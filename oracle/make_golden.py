#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE model (imported from /root/reference) on the
synthetic weights/frames of unet-_amd/synthetic.py.  Run in the build container only:

    python oracle/make_golden.py

/root/reference does not exist on the GPU box; only the resulting small fixtures travel.  The
reference module does `from torchvision import models` at import time (src/models/unetpp.py:9) and
torchvision is not installed; `models` is only dereferenced under `pretrained_encoder=True`
(unetpp.py:52-65, never taken by the north-star callers), so empty stub modules are registered first.
The frame-loop tail (softmax -> np.argmax -> uint8 -> class masks) is evaluated with the same torch /
numpy calls as infer_two_stage_burr.py:299-304 (that script itself imports cv2, which is absent).
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _load_synthetic():
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from unet_amd import synthetic
    return synthetic


def _import_reference():
    for name in ("torchvision", "torchvision.models"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.path.insert(0, REF)
    from src.models.unetpp import NestedUNet  # noqa
    return NestedUNet


def _import_rule_scripts():
    """The thresholded frame loops import cv2 (absent) at module level; their pure-NumPy helpers
    (softmax_np + the class rules) run fine with an empty cv2 stub.  Nothing else of them is executed
    (their main() sits under `if __name__ == '__main__'`)."""
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    import importlib
    if REF not in sys.path:
        sys.path.insert(0, REF)
    best = importlib.import_module("infer_video_3class_best")
    strict = importlib.import_module("infer_video_strict")
    fixed = importlib.import_module("infer_video_fixed")
    robust = importlib.import_module("infer_video_robust")
    geom = importlib.import_module("src.utils.geometry_enhanced")
    return best, strict, fixed, robust, geom


def rule_payload(mods, logits):
    """cable/tape masks of every rule family, with the reference's default parameters, per frame."""
    best, strict, fixed, robust, geom = mods
    out = {}
    names = (("thr", lambda p: best.thresholded_argmax(p)), ("thr_strict", lambda p: strict.thresholded_argmax_strict(p)),
             ("bgcheck", lambda p: fixed.strict_threshold_with_bg_check(p)), ("excl", lambda p: robust.exclusive_threshold(p)),
             ("excl_loose", lambda p: robust.exclusive_threshold(p, t_cable=0.34, t_tape=0.34, bg_margin=0.0, ct_margin=0.0)))
    probs_all = []
    for tag, fn in names:
        cs, ts = [], []
        for i in range(logits.shape[0]):
            probs = best.softmax_np(logits[i].transpose(1, 2, 0))          # infer_video_3class_best.py:197
            c, t = fn(probs)
            cs.append(c); ts.append(t)
            if tag == "thr":
                probs_all.append(probs)
        out[f"rule_{tag}_cable"] = np.stack(cs).astype(np.uint8)
        out[f"rule_{tag}_tape"] = np.stack(ts).astype(np.uint8)
    out["probs_hwc"] = np.stack(probs_all).astype(np.float32)
    # per-row widths of the plain argmax masks by the reference's own _compute_width_per_row (no smoothing: cv2)
    pred = np.argmax(np.stack(probs_all), axis=-1)
    out["rowwidth_cable"] = np.stack([geom._compute_width_per_row((pred[i] == 1).astype(np.uint8), smooth=False) for i in range(pred.shape[0])])
    out["rowwidth_tape"] = np.stack([geom._compute_width_per_row((pred[i] == 2).astype(np.uint8), smooth=False) for i in range(pred.shape[0])])
    return out


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def run_reference(NestedUNet, syn, C, ds, wseed, frames_u8, want_intermediates):
    sd_np = syn.make_state_dict(C, 3, ds, wseed)
    model = NestedUNet(num_classes=C, input_channels=3, deep_supervision=ds, pretrained_encoder=False)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
    model.eval()
    x = torch.from_numpy(syn.frames_to_chw_f32(frames_u8))
    inter = {}
    hooks = []
    if want_intermediates:
        names = {"conv0_0": "x0_0", "conv1_0": "x1_0", "conv2_0": "x2_0", "conv3_0": "x3_0", "conv4_0": "x4_0",
                 "conv3_1": "x3_1", "conv2_2": "x2_2", "conv1_3": "x1_3", "conv0_4": "x0_4"}
        for mod_name, t_name in names.items():
            hooks.append(getattr(model, mod_name).register_forward_hook(
                lambda m, i, o, t=t_name: inter.__setitem__(t, o.detach().clone().numpy())))
    with torch.no_grad():
        outputs = model(x)                                   # infer_two_stage_burr.py:294-297
        if isinstance(outputs, list):
            outputs = outputs[-1]
    probs = torch.softmax(outputs, dim=1).cpu().numpy()      # :299 (all frames, not just [0])
    pred = np.argmax(probs, axis=1).astype(np.uint8)         # :300
    for h in hooks:
        h.remove()
    return sd_np, outputs.numpy(), pred, inter


def margin_of(logits):
    s = np.sort(logits, axis=1)
    return s[:, -1] - s[:, -2]


def main():
    torch.manual_seed(0)
    os.makedirs(OUT, exist_ok=True)
    syn = _load_synthetic()
    NestedUNet = _import_reference()
    rule_mods = _import_rule_scripts()
    meta_lines = []

    # ---- small cases: full logits (+ every intermediate for the smallest one)
    small = [
        # tag, C, ds, wseed, B, H, W, kind, fseed, intermediates
        ("s_c3_32x32", 3, True, 2, 1, 32, 32, "smooth", 11, True),
        ("s_c3_64x64", 3, True, 8, 2, 64, 64, "uniform", 12, False),
        ("s_c7_48x80", 7, False, 2, 1, 48, 80, "smooth", 13, False),
        ("s_c3_128x96", 3, True, 2, 1, 128, 96, "smooth", 14, False),
    ]
    for tag, C, ds, wseed, B, H, W, kind, fseed, inter in small:
        frames = syn.make_frames_u8(B, H, W, kind, fseed)
        sd, logits, pred, t = run_reference(NestedUNet, syn, C, ds, wseed, frames, inter)
        payload = dict(num_classes=C, deep_supervision=ds, wseed=wseed, B=B, H=H, W=W, kind=kind, fseed=fseed,
                       frames_sha=sha(frames), weights_sha=sha(np.concatenate([v.ravel().astype(np.float64) for v in sd.values()])),
                       logits=logits.astype(np.float32), mask=pred,
                       mask_cable=(pred == 1).astype(np.uint8), mask_tape=(pred == 2).astype(np.uint8))
        for k, v in t.items():
            payload["t_" + k] = v.astype(np.float32)
        if C == 3 and tag != "s_c3_32x32":
            payload.update(rule_payload(rule_mods, logits))
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **payload)
        hist = np.bincount(pred.ravel(), minlength=C).tolist()
        meta_lines.append(f"{tag}: logits[{logits.min():.3f},{logits.max():.3f}] class_hist={hist} min_margin={margin_of(logits).min():.3e}")

    # ---- full-size cases: mask, subsampled logits, checksums, near-tie pixel list
    big = [
        ("b_c3_512x512", 3, True, 2, 2, 512, 512, ("smooth", "uniform"), 1234),
        ("b_c7_448x800", 7, False, 0, 1, 448, 800, ("smooth",), 1234),
    ]
    for tag, C, ds, wseed, B, H, W, kinds, fseed in big:
        frames = np.stack([syn.make_frame_u8(H, W, i, kinds[i % len(kinds)], fseed) for i in range(B)])
        sd, logits, pred, _ = run_reference(NestedUNet, syn, C, ds, wseed, frames, False)
        m = margin_of(logits)
        tie_idx = np.argwhere(m < 1e-3).astype(np.int32)              # (b, y, x) of near-tie pixels
        tie_margin = m[m < 1e-3].astype(np.float32)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"),
                            num_classes=C, deep_supervision=ds, wseed=wseed, B=B, H=H, W=W, kinds=np.array(kinds), fseed=fseed,
                            frames_sha=sha(frames), mask=pred, logits_sub8=logits[:, :, ::8, ::8].astype(np.float32),
                            logits_sha=sha(logits.astype(np.float32)), mask_sha=sha(pred),
                            tie_idx=tie_idx, tie_margin=tie_margin,
                            class_hist=np.stack([np.bincount(pred[i].ravel(), minlength=C) for i in range(B)]))
        meta_lines.append(f"{tag}: logits[{logits.min():.3f},{logits.max():.3f}] "
                          f"class_hist={np.bincount(pred.ravel(), minlength=C).tolist()} near_ties(<1e-3)={len(tie_margin)} "
                          f"min_margin={m.min():.3e}")

    # ---- BASELINE config 2 at its stated batch (16 frames, both frame kinds): masks, near-tie list, subsampled logits
    # and, for frames 0-1, every node x0_0 ... x0_4 sampled at seeded random positions (per-layer localisation at a
    # size where the deep layers span many 16x32 tiles)
    tag, C, ds, wseed, B, H, W, kinds, fseed = ("b_c3_512x512_b16", 3, True, 2, 16, 512, 512, ("smooth", "uniform"), 1234)
    frames = np.stack([syn.make_frame_u8(H, W, i, kinds[i % len(kinds)], fseed) for i in range(B)])
    sd, logits, pred, _ = run_reference(NestedUNet, syn, C, ds, wseed, frames, False)
    _, _, _, inter = run_reference(NestedUNet, syn, C, ds, wseed, frames[:2], True)
    m = margin_of(logits)
    payload = dict(num_classes=C, deep_supervision=ds, wseed=wseed, B=B, H=H, W=W, kinds=np.array(kinds), fseed=fseed,
                   frames_sha=sha(frames), mask=pred, logits_sub8=logits[:, :, ::8, ::8].astype(np.float32),
                   logits_sha=sha(logits.astype(np.float32)), mask_sha=sha(pred),
                   tie_idx=np.argwhere(m < 1e-3).astype(np.int32), tie_margin=m[m < 1e-3].astype(np.float32),
                   class_hist=np.stack([np.bincount(pred[i].ravel(), minlength=C) for i in range(B)]))
    prng = np.random.Generator(np.random.PCG64(99))
    for name, t in inter.items():
        lvl = int(name[1])
        h, w = H >> lvl, W >> lvl
        npos = 256 if lvl <= 1 else 64
        ys = prng.integers(0, h, npos).astype(np.int32); xs = prng.integers(0, w, npos).astype(np.int32)
        ys[:4] = (0, 0, h - 1, h - 1); xs[:4] = (0, w - 1, 0, w - 1)            # the four corners always
        payload["p_" + name] = np.stack([ys, xs], 1)
        payload["t_" + name] = t[:, :, ys, xs].astype(np.float32)              # [2, C, npos]
    np.savez_compressed(os.path.join(OUT, tag + ".npz"), **payload)
    meta_lines.append(f"{tag}: logits[{logits.min():.3f},{logits.max():.3f}] "
                      f"class_hist={np.bincount(pred.ravel(), minlength=C).tolist()} near_ties(<1e-3)={int((m < 1e-3).sum())} "
                      f"min_margin={m.min():.3e}")

    # ---- BASELINE config 5 (3-class 1024x1024): one frame through the reference, the same sparse payload
    tag, C, ds, wseed, B, H, W, kinds, fseed = ("b_c3_1024x1024", 3, True, 2, 1, 1024, 1024, ("smooth",), 1234)
    frames = np.stack([syn.make_frame_u8(H, W, 0, "smooth", fseed)])
    sd, logits, pred, _ = run_reference(NestedUNet, syn, C, ds, wseed, frames, False)
    m = margin_of(logits)
    np.savez_compressed(os.path.join(OUT, tag + ".npz"),
                        num_classes=C, deep_supervision=ds, wseed=wseed, B=B, H=H, W=W, kinds=np.array(kinds), fseed=fseed,
                        frames_sha=sha(frames), mask=pred, logits_sub8=logits[:, :, ::8, ::8].astype(np.float32),
                        logits_sha=sha(logits.astype(np.float32)), mask_sha=sha(pred),
                        tie_idx=np.argwhere(m < 1e-3).astype(np.int32), tie_margin=m[m < 1e-3].astype(np.float32))
    meta_lines.append(f"{tag}: logits[{logits.min():.3f},{logits.max():.3f}] "
                      f"class_hist={np.bincount(pred.ravel(), minlength=C).tolist()} near_ties(<1e-3)={int((m < 1e-3).sum())} "
                      f"min_margin={m.min():.3e}")

    # ---- SimpleUNet: the reference class on our synthetic state_dict
    from src.models.simple_unet import SimpleUNet
    simple_cases = [("su_c7_32x48", 7, 0, 1, 32, 48, "smooth", 31, True), ("su_c7_256x256", 7, 0, 1, 256, 256, "smooth", 32, False),
                    ("su_c3_64x40", 3, 1, 1, 64, 40, "uniform", 33, False)]
    for tag, C, wseed, B, H, W, kind, fseed, inter in simple_cases:
        sd_np = syn.make_simple_state_dict(C, 3, wseed)
        model = SimpleUNet(num_classes=C, num_channels=3)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd_np.items()}, strict=True)
        model.eval()
        frames = syn.make_frames_u8(B, H, W, kind, fseed)
        x = torch.from_numpy(syn.frames_to_chw_f32(frames))
        got = {}
        hooks = []
        if inter:
            for nm in ("enc1", "enc2", "enc3", "enc4", "dec3", "dec2", "dec1"):
                hooks.append(getattr(model, nm)[3].register_forward_hook(
                    lambda m, i, o, t=nm: got.__setitem__(t, o.detach().clone().numpy())))
        with torch.no_grad():
            out = model(x)
            probs = torch.softmax(out, dim=1).numpy()           # infer_video_simple.py:96
        for hk in hooks:
            hk.remove()
        logits = out.numpy()
        payload = dict(num_classes=C, wseed=wseed, B=B, H=H, W=W, kind=kind, fseed=fseed, frames_sha=sha(frames))
        if H * W <= 64 * 64:
            payload.update(logits=logits.astype(np.float32), probs=probs.astype(np.float32))
        else:
            m = margin_of(logits)
            payload.update(logits_sub4=logits[:, :, ::4, ::4].astype(np.float32), probs_sub4=probs[:, :, ::4, ::4].astype(np.float32),
                           tie_idx=np.argwhere(m < 1e-3).astype(np.int32))
        payload["mask"] = np.argmax(probs, axis=1).astype(np.uint8)
        for k, v in got.items():
            payload["t_" + k] = v.astype(np.float32)
        np.savez_compressed(os.path.join(OUT, tag + ".npz"), **payload)
        meta_lines.append(f"{tag}: logits[{logits.min():.3f},{logits.max():.3f}] class_hist={np.bincount(payload['mask'].ravel(), minlength=C).tolist()}")
    simple_manifest = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in SimpleUNet(7, 3).state_dict().items()]

    # ---- state_dict manifests (loader tests) for the two constructor variants the callers use
    manifest = {}
    for C, ds in ((3, True), (7, False)):
        model = NestedUNet(num_classes=C, input_channels=3, deep_supervision=ds)
        manifest[f"c{C}_ds{int(ds)}"] = [[k, list(v.shape), str(v.dtype).replace("torch.", "")]
                                         for k, v in model.state_dict().items()]
    manifest["simple_c7"] = simple_manifest
    import json
    with open(os.path.join(OUT, "state_dict_manifest.json"), "w") as f:
        json.dump(manifest, f, indent=0)

    # ---- ROI mapping of the frame loop (pure Python in the reference; cv2 only stubbed for the import)
    import importlib
    burr = importlib.import_module("infer_two_stage_burr")
    sizes = [(1920, 1080), (1080, 1920), (1280, 720), (640, 480), (512, 512), (800, 448), (3840, 2160), (1000, 333),
             (123, 457), (2048, 1536)]
    roi_cases = [{"size": list(sz), "target": [512, 512], "roi": list(burr.map_roi_to_original(sz, (512, 512)))}
                 for sz in sizes]
    roi_cases.append({"size": [1920, 1080], "target": [256, 256], "roi": list(burr.map_roi_to_original((1920, 1080), (256, 256)))})
    with open(os.path.join(OUT, "roi_map.json"), "w") as f:
        json.dump({"fixed_roi_512": burr.FIXED_ROI_512, "cases": roi_cases}, f, indent=0)
    with open(os.path.join(OUT, "README.txt"), "w") as f:
        f.write("Golden vectors produced by oracle/make_golden.py from the reference NestedUNet\n"
                f"(torch {torch.__version__}, {torch.get_num_threads()} threads, CPU fp32) on synthetic weights/frames.\n\n")
        f.write("\n".join(meta_lines) + "\n")
    print("\n".join(meta_lines))


if __name__ == "__main__":
    main()
