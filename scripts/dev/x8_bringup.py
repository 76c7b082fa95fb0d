"""Bring-up check of precision='exact8' on the GPU box: logits and every node against the committed golden fixtures
(reference outputs), next to 'exact' and 'fast' on the same inputs.   python scripts/dev/x8_bringup.py"""
import os, sys
import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_golden                       # noqa: E402
from unet_amd import synthetic as syn                  # noqa: E402
from unet_amd.nested_unet import NestedUNet            # noqa: E402
import unetpp_oracle as oracle                         # noqa: E402

NODES = ("x0_0", "x1_0", "x2_0", "x3_0", "x4_0", "x3_1", "x2_2", "x1_3", "x0_4")
for tag in ("s_c3_32x32", "s_c3_64x64", "s_c7_48x80", "s_c3_128x96"):
    g = load_golden(tag)
    B, H, W, C = int(g["B"]), int(g["H"]), int(g["W"]), int(g["num_classes"])
    frames = syn.make_frames_u8(B, H, W, str(g["kind"]), int(g["fseed"]))
    x = torch.from_numpy(syn.frames_to_chw_f32(frames)).cuda()
    sd = syn.make_state_dict(C, 3, bool(g["deep_supervision"]), int(g["wseed"]))
    for prec in ("exact", "exact8", "fast"):
        m = NestedUNet(C, deep_supervision=bool(g["deep_supervision"]), precision=prec, max_batch=B, max_hw=(H, W)).to("cuda:0")
        m.load_state_dict(sd, strict=True); m.eval()
        mask, logits = m.segment(x, return_logits=True)
        torch.cuda.synchronize()
        lg = logits.cpu().numpy()
        err = float(np.abs(lg - g["logits"]).max())
        flips = int((mask.cpu().numpy() != g["mask"]).sum())
        line = f"{tag} {prec:7s} max|dlogit| {err:.3e} flips {flips} status {m.status()}"
        if tag == "s_c3_32x32":
            m.debug_keep_intermediates(True)
            m(x); torch.cuda.synchronize()
            per = []
            for name in NODES:
                got = m.debug_activation(name, B, H, W)
                ref = g["t_" + name]
                per.append(f"{name}:{float(np.abs(got - ref).max() / max(1e-9, np.abs(ref).max())):.1e}")
            line += "  rel-to-max node errors " + " ".join(per)
        print(line, flush=True)
        del m
