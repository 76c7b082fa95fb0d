#!/usr/bin/env python3
"""bench.py — frames/s of 512x512 3-class UNet++ inference on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic frames: device-resident float32
[B,3,512,512] input -> device-resident uint8 [B,512,512] class mask (model call + argmax of
infer_two_stage_burr.py:294-300).  B = 16 frames per GPU (BASELINE config 2); with N GPUs every rank
runs its own 16-frame shard (weak scaling, config 3 = 8 x 16) with weights broadcast from rank 0 over
RCCL and no steady-state collective.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GFLOP_PER_FRAME = {(3, 512, 512): 111.636}
MFMA_F16_PEAK_TFLOPS = 2500.0                       # dense fp16 MFMA, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def algorithmic_gflop(C, H, W):
    nb = (32, 64, 128, 256, 512)
    macs = 0
    for l in range(5):
        px = (H >> l) * (W >> l)
        cin = 3 if l == 0 else nb[l - 1]
        macs += px * 9 * (cin * nb[l] + nb[l] * nb[l])
    for l in range(4):
        px = (H >> l) * (W >> l)
        macs += px * 9 * ((nb[l] + nb[l + 1]) * nb[l] + nb[l] * nb[l])
    macs += H * W * 32 * C
    return 2.0 * macs / 1e9


def cpu_baseline(sd, syn, C, H, W, n_frames, gpu_model, torch):
    """The oracle's torch-CPU restatement (what the reference's --device cpu path executes), batch 1
    per call like the reference frame loop, timed on this host's cores; the same frames go through
    the GPU engine and the two masks are compared."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import unetpp_oracle as oracle
    frames = syn.make_frames_u8(n_frames, H, W, "smooth", 4321)
    x = syn.frames_to_chw_f32(frames)
    # pick the thread count that serves this host best (the default of one thread per logical CPU
    # oversubscribes a 256-CPU box on a batch-1 conv net): 2 frames per candidate, then the timed sample
    default_threads = torch.get_num_threads()
    trials = {}
    for nt in sorted({8, 16, 32, 64, default_threads}):
        if nt > default_threads:
            continue
        torch.set_num_threads(nt)
        oracle.torch_segment(sd, x[:1])                                 # warm
        t0 = time.perf_counter()
        for i in range(2):
            oracle.torch_segment(sd, x[i:i + 1])
        trials[nt] = 2 / (time.perf_counter() - t0)
    best_nt = max(trials, key=trials.get)
    torch.set_num_threads(best_nt)
    oracle.torch_segment(sd, x[:1])
    t0 = time.perf_counter()
    ref_logits = [oracle.torch_forward(sd, x[i:i + 1]) for i in range(n_frames)]
    ref_masks = [oracle.masks_from_logits(l)[0] for l in ref_logits]
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    ref_logits = np.concatenate(ref_logits); ref_masks = np.concatenate(ref_masks)
    mask, logits = gpu_model.segment(torch.from_numpy(x).cuda(), return_logits=True)
    torch.cuda.synchronize()
    err = float(np.abs(logits.cpu().numpy() - ref_logits).max())
    flips = mask.cpu().numpy() != ref_masks
    margin = oracle.top2_margin(ref_logits)
    base = {"value": n_frames / dt, "unit": "frames/s", "cores": int(best_nt), "kind": "port",
            "sample": f"{n_frames} frames of {C}-class {H}x{W}, batch 1 per call, torch {torch.__version__} CPU fp32 "
                      f"(oracle/unetpp_oracle.py torch_forward + softmax/argmax), host cpu_count={os.cpu_count()}, "
                      f"threads tried (frames/s): " + ", ".join(f"{k}:{v:.2f}" for k, v in sorted(trials.items()))}
    parity = {"frames": n_frames, "max_abs_logit_err": err, "mask_flips": int(flips.sum()),
              "mask_pixels": int(flips.size),
              "flips_outside_near_ties": int((flips & (margin > 2 * err + 1e-7)).sum()), "logit_tol": 1e-3}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--precision", default="exact", choices=["exact", "fast"])
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-fast-leg", action="store_true")
    ap.add_argument("--no-e2e-leg", action="store_true", help="skip the PCIe-inclusive informational leg")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from unet_amd import sharding, synthetic as syn
    from unet_amd.nested_unet import NestedUNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    # one process per GPU; UNETPP_BENCH_SHARE_GPU=1 maps every rank onto the visible GPUs round-robin
    # (rehearsal of the N>1 control flow on a 1-GPU box, together with UNETPP_DIST_BACKEND=gloo)
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if os.environ.get("UNETPP_BENCH_SHARE_GPU") else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device(f"cuda:{dev_index}")
    backend = os.environ.get("UNETPP_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    C, H, W, B = args.classes, args.height, args.width, args.batch
    ds = C == 3
    sd = syn.make_state_dict(C, 3, ds, 2) if rank == 0 else None

    def make_model(precision):
        m = NestedUNet(C, deep_supervision=ds, precision=precision, max_batch=B, max_hw=(H, W),
                       micro_batch=args.micro_batch, streams=args.streams).to(dev)
        if world > 1:
            m._ensure_engine(B, H, W)
            sharding.load_replicated(m, sd, C)           # RCCL broadcast of the weight blob from rank 0
        else:
            m.load_state_dict(sd, strict=True)
        return m.eval()

    # per-rank shard of the global batch: frames [rank*B, (rank+1)*B)
    lo, hi = sharding.shard_range(B * world, rank, world)
    frames = syn.make_frames_u8(hi - lo, H, W, "smooth", 1234, first=lo)
    x = torch.from_numpy(syn.frames_to_chw_f32(frames)).to(dev)

    def timed(model, steps, warmup, profile):
        for _ in range(warmup):
            model.segment(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if profile:
            model.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            model.segment(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        recs = model.profile_read() if profile else []
        if profile:
            model.profile(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, recs

    model = make_model(args.precision)
    dt, recs = timed(model, args.steps, args.warmup, profile=True)
    fps = B * world * args.steps / dt

    # ---- roofline of the dominant kernel (HIP events recorded on the launch stream inside the timed region)
    agg = {}
    for name, ms, fl, by in recs:
        k = name.split("|")[-1]
        a = agg.setdefault(k, [0.0, 0.0, 0.0, 0])
        a[0] += ms; a[1] += fl; a[2] += by; a[3] += 1
    dom = max(agg.items(), key=lambda kv: kv[1][0]) if agg else None
    roofline = None
    pmc = {}
    try:   # HBM traffic per launch from the committed rocprofv3 --pmc passes (profiles/README.md), if present
        import glob
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))
        if files:
            pmc = json.load(open(files[-1]))["kernels"]
            pmc_src = os.path.basename(files[-1])
    except Exception:
        pmc = {}
    if dom:
        k, (ms, fl, by, cnt) = dom
        tf = fl / (ms * 1e-3) / 1e12
        mfma_mult = 3.0 if args.precision == "exact" else 1.0      # exact mode issues 3 MFMAs per product
        traffic = pmc[k]["hbm_bytes_per_launch"] if (k in pmc and args.precision == "exact" and (C, H, W, B) == (3, 512, 512, 16)) else None
        roofline = {"bound": "mfma", "kernel": k, "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": tf / MFMA_F16_PEAK_TFLOPS, "traffic": traffic,
                    "traffic_source": (f"profiles/{pmc_src}: (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes"
                                       if traffic is not None else None),
                    "algorithmic_bytes_per_launch": by / cnt,
                    "mfma_issued_tflops": tf * mfma_mult, "mfma_issued_frac": tf * mfma_mult / MFMA_F16_PEAK_TFLOPS,
                    "avg_launch_ms": ms / cnt, "launches": cnt,
                    "algorithmic_gflop_per_launch": fl / cnt / 1e9,
                    "algorithmic_hbm_gbs": by / (ms * 1e-3) / 1e9,
                    "time_share": ms / sum(v[0] for v in agg.values())}

    out = {
        "metric": "frames/sec 512x512 3-class UNet++ inference; mask vs CPU reference" if (C, H, W) == (3, 512, 512)
                  else f"frames/sec {H}x{W} {C}-class UNet++ inference",
        "value": fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.precision == "fast" else "f16 (MFMA operands split hi+lo, 3 MFMAs per product, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"UNet++ {C}-class {H}x{W} batch={B}/GPU fp16-MFMA on {world} MI355X, synthetic frames, "
                               f"f32 NCHW in HBM -> uint8 mask in HBM",
                   "precision": args.precision, "frames_per_gpu": B, "global_batch": B * world,
                   "micro_batch": args.micro_batch or B, "parallelism": f"frame-sharded x{world}, weights replicated (RCCL bcast)"},
        "whole_net": {"gflop_per_frame": algorithmic_gflop(C, H, W),
                      "achieved_tflops": fps * algorithmic_gflop(C, H, W) / 1e3,
                      "frac_of_f16_mfma_peak": fps * algorithmic_gflop(C, H, W) / 1e3 / (MFMA_F16_PEAK_TFLOPS * world)},
        "roofline": roofline,
    }
    if rank == 0:
        out["kernels"] = {k: {"ms_per_step": v[0] / args.steps, "tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[0] else 0.0,
                              "alg_gbs": (v[2] / (v[0] * 1e-3) / 1e9) if v[0] else 0.0, "launches_per_step": v[3] / args.steps}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}

    # ---- CPU baseline + parity of the measured mode on the same frames (rank 0, N=1 only)
    if rank == 0 and world == 1 and args.cpu_frames > 0:
        base, parity = cpu_baseline(sd, syn, C, H, W, args.cpu_frames, model, torch)
        out["cpu_baseline"] = base
        out["parity"] = parity
    else:
        out["cpu_baseline"] = None

    # ---- informational: the PCIe-inclusive rate (never `value`): pinned uint8 BGR frames -> H2D -> engine (BGR->RGB,
    # /255 fused) -> uint8 masks -> D2H into pinned memory, two engines on two streams so copies overlap compute
    if not args.no_e2e_leg and world == 1:
        ma, mb = model, make_model(args.precision)
        eng = [ma, mb]
        st = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
        h_in = [torch.from_numpy(np.ascontiguousarray(frames)).pin_memory() for _ in range(2)]
        h_out = [torch.empty((B, H, W), dtype=torch.uint8).pin_memory() for _ in range(2)]
        d_in = [torch.empty((B, H, W, 3), dtype=torch.uint8, device=dev) for _ in range(2)]

        def e2e_step(i):
            k = i & 1
            with torch.cuda.stream(st[k]):
                d_in[k].copy_(h_in[k], non_blocking=True)                                  # H2D (pinned, contiguous)
                h_out[k].copy_(eng[k].segment(d_in[k]), non_blocking=True)               # D2H
        for i in range(max(2, args.warmup)):
            e2e_step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            e2e_step(i)
        torch.cuda.synchronize()
        dte = time.perf_counter() - t0
        out["end_to_end"] = {"value": B * args.steps / dte, "unit": "frames/s", "ms_per_step": dte / args.steps * 1e3,
                             "path": "pinned uint8 BGR frames (0.79 MB/frame) H2D -> engine -> uint8 masks (0.26 MB/frame) D2H, "
                                     "2 engines on 2 streams; informational, not `value`",
                             "mask_equals_resident_path": bool(torch.equal(h_out[0], ma.segment(x).cpu()))}
        del mb, eng, d_in

    # ---- informational second leg: the other precision mode on the same workload
    if not args.no_fast_leg and world == 1:
        other = "fast" if args.precision == "exact" else "exact"
        del model
        m2 = make_model(other)
        dt2, _ = timed(m2, args.steps, args.warmup, profile=False)
        leg = {"precision": other, "value": B * args.steps / dt2, "unit": "frames/s", "ms_per_step": dt2 / args.steps * 1e3}
        if rank == 0 and args.cpu_frames > 0:
            _, p2 = cpu_baseline(sd, syn, C, H, W, min(args.cpu_frames, 2), m2, torch)
            leg["parity"] = p2
        out["other_precision"] = leg

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
