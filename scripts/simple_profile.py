#!/usr/bin/env python3
"""Per-launch timing of SimpleUNet (7-class 256x256, the shape of infer_video_simple.py:88)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import synthetic as syn
from unet_amd.nested_unet import SimpleUNet
prec = sys.argv[1] if len(sys.argv) > 1 else "exact"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
m = SimpleUNet(7, precision=prec, max_batch=B, max_hw=(H, W)).to("cuda:0")
m.load_state_dict(syn.make_simple_state_dict(7, 3, 0))
x = torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(B, H, W, "smooth", 1))).cuda()
for _ in range(3): m.predict_proba(x)
torch.cuda.synchronize()
m.profile(True)
reps = 10
for _ in range(reps): m.predict_proba(x)
torch.cuda.synchronize()
recs = m.profile_read(); n = len(recs) // reps; tot = 0
for i in range(n):
    ms = sum(recs[i + r * n][1] for r in range(reps)) / reps; tot += ms
    print(f"{recs[i][0]:60s} {ms*1e3:8.1f} us {recs[i][2]/ms/1e9:8.1f} TF/s {recs[i][3]/ms/1e6:8.1f} GB/s")
print(f"SimpleUNet {prec} B={B} {H}x{W}: sum {tot:.3f} ms -> {B/tot*1e3:.0f} frames/s")
