#!/usr/bin/env python3
"""Soak: repeated forwards of several configurations must reproduce their first result bit for bit (a race in the
loader / consumer hand-overs, or a stale read in the split-K hand-off between workgroups, would show up as a flipped bit
sooner or later).  Batch 1 and 2 at 512x512 take split-K plans at levels 3-4 (conv3x3_ws.h).
usage: python scripts/soak_determinism.py [reps] [exact|exact8|fast]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unet_amd import synthetic as syn
from unet_amd.nested_unet import NestedUNet

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
prec = sys.argv[2] if len(sys.argv) > 2 else "exact"
cfgs = [(3, 16, 512, 512), (7, 8, 448, 800), (3, 2, 1024, 1024), (3, 1, 512, 512), (3, 2, 512, 512), (7, 1, 448, 800), (3, 5, 80, 112),
        (3, 3, 48, 176), (5, 4, 256, 320)]
bad = 0
for C, B, H, W in cfgs:
    m = NestedUNet(C, deep_supervision=(C == 3), precision=prec, max_batch=B, max_hw=(H, W)).to("cuda:0")
    m.load_state_dict(syn.make_state_dict(C, 3, C == 3, 2))
    x = torch.from_numpy(syn.frames_to_chw_f32(syn.make_frames_u8(B, H, W, "smooth", 99))).cuda()
    mask0, log0 = m.segment(x, return_logits=True)
    mask0, log0 = mask0.clone(), log0.clone()
    t0 = time.time()
    for i in range(reps):
        mask, log = m.segment(x, return_logits=True)
        if not (torch.equal(mask, mask0) and torch.equal(log, log0)):
            bad += 1
            print(f"MISMATCH C={C} B={B} {H}x{W} rep {i}: {(log != log0).sum().item()} logits differ", flush=True)
    torch.cuda.synchronize()
    print(f"{prec} C={C} B={B} {H}x{W}: {reps} repeats identical={bad == 0} status={m.status()} ({time.time() - t0:.1f} s)", flush=True)
    del m
print("soak", "FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
