#!/usr/bin/env python3
"""Per-launch timing table of one forward (HIP events from the engine's profiling hooks).
usage: python scripts/layer_profile.py [exact|fast] [B] [H] [W] [C] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import synthetic as syn
from unet_amd.nested_unet import NestedUNet

prec = sys.argv[1] if len(sys.argv) > 1 else "exact"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 512
W = int(sys.argv[4]) if len(sys.argv) > 4 else 512
C = int(sys.argv[5]) if len(sys.argv) > 5 else 3
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
mb = int(sys.argv[7]) if len(sys.argv) > 7 else 0
ns = int(sys.argv[8]) if len(sys.argv) > 8 else 1
m = NestedUNet(C, deep_supervision=(C == 3), precision=prec, max_batch=B, max_hw=(H, W), micro_batch=mb, streams=ns).to("cuda:0")
m.load_state_dict(syn.make_state_dict(C, 3, C == 3, 2))
x = torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(B, H, W, "smooth", 1234))).cuda()
for _ in range(3):
    m.segment(x)
torch.cuda.synchronize()
m.profile(True)
for _ in range(reps):
    m.segment(x)
torch.cuda.synchronize()
recs = m.profile_read()
n = len(recs) // reps
tot = 0.0
print(f"{prec} B={B} {H}x{W} C={C}")
for i in range(n):
    ms = sum(recs[i + r * n][1] for r in range(reps)) / reps
    name, _, fl, by = recs[i]
    tot += ms
    print(f"{name:58s} {ms*1e3:8.1f} us  {fl/ms/1e9:8.1f} TF/s  {by/ms/1e6:8.1f} GB/s")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    m.segment(x)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / reps
print(f"wall {wall*1e3:.3f} ms -> {B/wall:.0f} frames/s (mb={mb}, streams={ns})")
print(f"sum {tot:.3f} ms  -> {B/tot*1e3:.0f} frames/s (sum of kernels)")
