"""Bring-up check of precision='exact8' for SimpleUNet on the GPU box: logits and the nodes the small fixtures hold against
the committed golden fixtures (reference outputs), next to 'exact' and 'fast'.   python scripts/dev/x8_simple_bringup.py"""
import os, sys
import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_golden                       # noqa: E402
from unet_amd import synthetic as syn                  # noqa: E402
from unet_amd.nested_unet import SimpleUNet            # noqa: E402

for tag in ("su_c7_32x48", "su_c3_64x40", "su_c7_256x256"):
    g = load_golden(tag)
    B, H, W, C = int(g["B"]), int(g["H"]), int(g["W"]), int(g["num_classes"])
    frames = syn.make_frames_u8(B, H, W, str(g["kind"]), int(g["fseed"]))
    x = torch.from_numpy(syn.frames_to_chw_f32(frames)).cuda()
    sd = syn.make_simple_state_dict(C, 3, int(g["wseed"]))
    for prec in ("exact", "exact8", "fast"):
        m = SimpleUNet(num_classes=C, num_channels=3, precision=prec, max_batch=B, max_hw=(H, W)).to("cuda:0")
        m.load_state_dict(sd, strict=True); m.eval()
        logits = m(x)
        torch.cuda.synchronize()
        lg = logits.cpu().numpy()
        if "logits" in g.files:
            err = float(np.abs(lg - g["logits"]).max())
        else:
            err = float(np.abs(lg[:, :, ::4, ::4] - g["logits_sub4"]).max())
        line = f"{tag} {prec:7s} max|dlogit| {err:.3e} status {m.status()}"
        per = []
        for k in [f for f in g.files if f.startswith("t_")]:
            got = m.debug_activation(k[2:], B, H, W)
            per.append(f"{k[2:]}:{float(np.abs(got - g[k]).max() / max(1e-9, np.abs(g[k]).max())):.1e}")
        print(line + ("  rel-to-max node errors " + " ".join(per) if per else ""), flush=True)
        del m
