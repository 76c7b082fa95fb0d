#!/usr/bin/env python3
"""bench.py — frames/s of 512x512 3-class UNet++ inference on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic frames: device-resident float32
[B,3,512,512] input -> device-resident uint8 [B,512,512] class mask (model call + argmax of
infer_two_stage_burr.py:294-300).  B = 16 frames per GPU (BASELINE config 2); with N GPUs every rank
runs its own 16-frame shard (weak scaling, config 3 = 8 x 16) with weights broadcast from rank 0 over
RCCL and no steady-state collective.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus 8                      # starts its own 8 ranks (one fresh process per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W    # or under an external launcher (RANK/WORLD_SIZE in the env)

Other workloads of SURVEY §8(d) come from the same script: `--classes 7 --height 448 --width 800 --batch 32`
(config 4), `--height 1024 --width 1024 --batch 8` (config 5), `--arch simple --classes 7 --height 256 --width 256`
(SimpleUNet, §8(f) row 3).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DTYPES = {"exact": "f16 (MFMA operands split hi+lo, 3 MFMAs per product, f32 accumulate)",
          "exact8": "f16 + fp8 (hi*hi in fp16 MFMA; lo*w and x*w_lo from e5m2 / e4m3 operands in one block-scaled K=64 MFMA per tap "
                    "pair; f32 accumulate)",
          "fast": "f16"}
MFMA_F16_PEAK_TFLOPS = 2500.0                       # dense fp16 MFMA, MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def algorithmic_gflop(arch, C, H, W):
    """2 x MACs of one frame (111.636 for the 3-class 512x512 NestedUNet)."""
    macs = 0
    if arch == "nested":
        nb = (32, 64, 128, 256, 512)
        for l in range(5):
            px = (H >> l) * (W >> l)
            cin = 3 if l == 0 else nb[l - 1]
            macs += px * 9 * (cin * nb[l] + nb[l] * nb[l])
        for l in range(4):
            px = (H >> l) * (W >> l)
            macs += px * 9 * ((nb[l] + nb[l + 1]) * nb[l] + nb[l] * nb[l])
        macs += H * W * 32 * C
    else:                                            # SimpleUNet, src/models/simple_unet.py:30-92
        sb = (64, 128, 256, 512)
        for l in range(4):
            px = (H >> l) * (W >> l)
            cin = 3 if l == 0 else sb[l - 1]
            macs += px * 9 * (cin * sb[l] + sb[l] * sb[l])
        for l in range(3):
            px = (H >> l) * (W >> l)
            macs += (px // 4) * 4 * sb[l + 1] * sb[l]                      # ConvTranspose2d k2 s2
            macs += px * 9 * (2 * sb[l] * sb[l] + sb[l] * sb[l])
        macs += H * W * 64 * C
    return 2.0 * macs / 1e9


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(args) -> int:
    """`python bench.py --gpus N` from a plain shell: start N fresh processes (one per GPU) and relay rank 0's JSON
    line.  The parent never touches the GPU (no HIP call, nothing re-exec'ed), so every rank initialises its own
    device from scratch; children get RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like under torch.distributed.run."""
    n = args.gpus
    share = bool(os.environ.get("UNETPP_BENCH_SHARE_GPU")) or args.no_engine
    if not share:
        # device_count() does not initialise HIP; asked in a child anyway so that this process stays GPU-free
        try:
            q = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                               capture_output=True, text=True, timeout=600)
            ndev = int(q.stdout.strip().splitlines()[-1]) if q.returncode == 0 and q.stdout.strip() else 0
        except (subprocess.SubprocessError, ValueError):
            ndev = 0
        if ndev < n:
            print(f"bench.py: --gpus {n} needs {n} HIP devices, this machine shows {ndev} "
                  f"(one process per GPU; set UNETPP_BENCH_SHARE_GPU=1 only to rehearse the control flow)", file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs = []
    # host threads per rank (torch's intra-op pool, only used by the CPU-side helpers): the host's cores shared out
    # among the ranks, unless the caller has set OMP_NUM_THREADS (N ranks x all cores each would oversubscribe the host)
    threads = str(max(1, min(16, (os.cpu_count() or n) // n)))
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), UNETPP_BENCH_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", threads)
        # rank 0's stdout carries the JSON line; the other ranks' stdout joins stderr so the parent prints one line
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # rank 0's pipe is drained while it runs: a library that writes more than a pipe buffer to stdout must not block it
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.extend(iter(procs[0].stdout.readline, "")), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("UNETPP_BENCH_TIMEOUT", "1500"))
    first_fail = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        bad = [c for c in codes if c not in (None, 0)]
        if bad and first_fail is None:
            first_fail = time.time()
        if (first_fail is not None and time.time() - first_fail > 15) or time.time() > deadline:
            for p in procs:                      # a rank died (or the job hangs): stop exactly the ranks we started
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        time.sleep(0.2)
    reader.join(timeout=10)
    out0 = "".join(chunks)
    codes = [p.poll() if p.poll() is not None else -9 for p in procs]
    for line in out0.splitlines():             # the JSON line goes to stdout, library chatter ("[Gloo] Rank 0 ...") to stderr
        print(line, file=sys.stdout if line.startswith("{") else sys.stderr, flush=True)
    worst = max((abs(c) for c in codes), default=0)
    if worst:
        print(f"bench.py: rank exit codes {codes}", file=sys.stderr)
        return worst if worst < 256 else 1
    return 0


# ------------------------------------------------------------------------------------------------ CPU baseline
def cpu_baseline(sd, syn, arch, C, H, W, n_frames, gpu_model, torch, ref=None):
    """The oracle's torch-CPU restatement (what the reference's --device cpu path executes), batch 1
    per call like the reference frame loop, timed on this host's cores; the same frames go through
    the GPU engine and the two masks are compared."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import unetpp_oracle as oracle
    fwd = oracle.torch_forward if arch == "nested" else oracle.simple_unet_torch_forward
    frames = syn.make_frames_u8(n_frames, H, W, "smooth", 4321)
    x = syn.frames_to_chw_f32(frames)
    if ref is not None and ref.get("n", 0) >= n_frames:        # the CPU side of these frames was computed (and timed) for the headline leg
        ref_logits, ref_masks = ref["logits"][:n_frames], ref["masks"][:n_frames]
        mask, logits = gpu_model.segment(torch.from_numpy(x).cuda(), return_logits=True)
        torch.cuda.synchronize()
        return None, _parity(oracle, logits.cpu().numpy(), mask.cpu().numpy(), ref_logits, ref_masks, n_frames, gpu_model)

    def segment(xi):
        lg = fwd(sd, xi)
        return lg, oracle.masks_from_logits(lg)[0]
    # pick the thread count that serves this host best (the default of one thread per logical CPU
    # oversubscribes a 256-CPU box on a batch-1 conv net): 2 frames per candidate, then the timed sample
    default_threads = torch.get_num_threads()
    trials = {}
    for nt in sorted({8, 16, 32, 64, default_threads}):
        if nt > default_threads:
            continue
        torch.set_num_threads(nt)
        segment(x[:1])                                                   # warm
        t0 = time.perf_counter()
        for i in range(2):
            segment(x[i % n_frames:i % n_frames + 1])
        trials[nt] = 2 / (time.perf_counter() - t0)
    best_nt = max(trials, key=trials.get)
    torch.set_num_threads(best_nt)
    segment(x[:1])
    t0 = time.perf_counter()
    res = [segment(x[i:i + 1]) for i in range(n_frames)]
    dt = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    ref_logits = np.concatenate([r[0] for r in res]); ref_masks = np.concatenate([r[1] for r in res])
    if ref is not None:
        ref.update(n=n_frames, logits=ref_logits, masks=ref_masks)
    mask, logits = gpu_model.segment(torch.from_numpy(x).cuda(), return_logits=True)
    torch.cuda.synchronize()
    base = {"value": n_frames / dt, "unit": "frames/s", "cores": int(best_nt), "kind": "port",
            "sample": f"{n_frames} frames of {C}-class {H}x{W}, batch 1 per call, torch {torch.__version__} CPU fp32 "
                      f"(oracle/unetpp_oracle.py {fwd.__name__} + softmax/argmax), host cpu_count={os.cpu_count()}, "
                      f"threads tried (frames/s): " + ", ".join(f"{k}:{v:.2f}" for k, v in sorted(trials.items()))}
    return base, _parity(oracle, logits.cpu().numpy(), mask.cpu().numpy(), ref_logits, ref_masks, n_frames, gpu_model)


def _parity(oracle, logits, mask, ref_logits, ref_masks, n_frames, gpu_model):
    import numpy as np
    err = float(np.abs(logits - ref_logits).max())
    flips = mask != ref_masks
    margin = oracle.top2_margin(ref_logits)
    return {"frames": n_frames, "max_abs_logit_err": err, "mask_flips": int(flips.sum()),
            "mask_pixels": int(flips.size), "largest_margin_at_a_flip": float(margin[flips].max()) if flips.any() else 0.0,
            "flips_outside_near_ties": int((flips & (margin > 2 * err + 1e-7)).sum()), "logit_tol": 1e-3,
            "range_status": int(gpu_model.status())}


def measured_traffic(kernel, workload_key):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 --pmc passes, but only from a file that was
    recorded for THIS build (same source hash) and workload; otherwise None plus the reason."""
    import glob
    from unet_amd import _lib
    want = _lib.source_hash()
    seen = []
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except (OSError, ValueError):
            continue
        seen.append(f"{os.path.basename(path)}:{d.get('src_hash', 'untagged')}")
        if d.get("src_hash") != want or d.get("workload") != workload_key:
            continue
        if kernel not in d.get("kernels", {}):
            return None, f"profiles/{os.path.basename(path)} matches this build but has no kernel {kernel!r}"
        return d["kernels"][kernel]["hbm_bytes_per_launch"], (
            f"profiles/{os.path.basename(path)} (src:{want}): (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, separate --pmc passes")
    return None, f"no profiles/*_pmc_traffic.json recorded for build src:{want}, workload {workload_key} (have: {', '.join(seen[:4]) or 'none'})"


class _NoEngine:
    """Stand-in for the engine in `--no-engine` runs (CPU rehearsal of the launcher / collective control flow under
    tests/test_bench_launcher.py).  It computes nothing; lines produced with it say so and carry value = null."""
    def __init__(self, B, H, W):
        import torch
        self._mask = torch.zeros((B, H, W), dtype=torch.uint8)

    def segment(self, x, return_logits=False):
        time.sleep(0.002)
        return self._mask

    def status(self):
        return 0


# ------------------------------------------------------------------------------------------------ one rank
def run_rank(args) -> int:
    import numpy as np
    import torch
    import torch.distributed as dist
    from unet_amd import _lib, sharding, synthetic as syn
    from unet_amd.nested_unet import NestedUNet, SimpleUNet

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} in the environment", file=sys.stderr)
        return 2
    dry = args.no_engine
    if dry and os.environ.get("UNETPP_BENCH_FAIL_RANK") == str(rank):     # launcher test: a rank that dies at start-up
        print(f"bench.py: rank {rank} failing on request (UNETPP_BENCH_FAIL_RANK)", file=sys.stderr)
        return 3
    backend = os.environ.get("UNETPP_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
    if dry:
        dev = torch.device("cpu"); dev_index = -1
        backend = "gloo"
    else:
        if not torch.cuda.is_available():
            print("bench.py needs a HIP device: the engine has no CPU fallback", file=sys.stderr)
            return 2
        # one process per GPU; UNETPP_BENCH_SHARE_GPU=1 maps every rank onto the visible GPUs round-robin
        # (rehearsal of the N>1 control flow on a 1-GPU box, together with UNETPP_DIST_BACKEND=gloo)
        ndev = torch.cuda.device_count()
        if local_rank >= ndev and not os.environ.get("UNETPP_BENCH_SHARE_GPU"):
            print(f"bench.py: rank {rank} has no GPU of its own ({ndev} visible)", file=sys.stderr)
            return 2
        dev_index = local_rank % ndev
        torch.cuda.set_device(dev_index)
        dev = torch.device(f"cuda:{dev_index}")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    comm_dev = dev if backend == "nccl" else torch.device("cpu")

    def sync():
        if not dry:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(v):
        if world == 1:
            return float(v)
        t = torch.tensor([v], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def gather(obj):
        if world == 1:
            return [obj]
        out = [None] * world
        dist.all_gather_object(out, obj)
        return out

    arch, C, H, W, B = args.arch, args.classes, args.height, args.width, args.batch
    ds = C == 3
    if arch == "nested":
        sd = syn.make_state_dict(C, 3, ds, 2) if rank == 0 else None
    else:
        sd = syn.make_simple_state_dict(C, 3, 0) if rank == 0 else None
    bcast = {"bytes": 0, "ms": None}

    def make_model(precision, timed_bcast=False):
        if dry:
            return _NoEngine(B, H, W)
        if arch == "nested":
            m = NestedUNet(C, deep_supervision=ds, precision=precision, max_batch=B, max_hw=(H, W),
                           micro_batch=args.micro_batch, streams=args.streams).to(dev)
        else:
            m = SimpleUNet(C, 3, precision=precision, max_batch=B, max_hw=(H, W), micro_batch=args.micro_batch,
                           streams=args.streams).to(dev)
        if world > 1:
            m._ensure_engine(B, H, W)
            barrier(); sync()                                   # the communicator exists before the timed broadcast
            t0 = time.perf_counter()
            blob = sharding.load_replicated(m, sd)              # RCCL broadcast of the weight blob from rank 0
            sync()
            if timed_bcast:
                bcast["bytes"] = int(blob.numel()); bcast["ms"] = (time.perf_counter() - t0) * 1e3
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
        sync(); barrier(); sync()
        if profile:
            model.profile(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            model.segment(x)
        sync()
        own = time.perf_counter() - t0                  # this rank alone, before waiting for the others
        barrier(); sync()
        dt = time.perf_counter() - t0
        recs = model.profile_read() if profile else []
        if profile:
            model.profile(False)
        return max_over_ranks(dt), own, recs

    if world > 1:
        barrier()                                       # first collective: sets the communicator up, untimed
    model = make_model(args.precision, timed_bcast=True)
    # `value`: no per-launch events in the timed region.  The timed loop (W warm-up + exactly K steps between barriers) runs
    # three times; the median sample is reported (boxes of the pool and even consecutive loops differ by a few percent),
    # all three are printed.
    samples = [timed(model, args.steps, args.warmup if i == 0 else 1, profile=False) for i in range(1 if dry else 3)]
    dt, own_dt, _ = sorted(samples, key=lambda t: t[0])[len(samples) // 2]
    fps = B * world * args.steps / dt
    gflop = algorithmic_gflop(arch, C, H, W)

    # ---- roofline of the dominant kernel: a second, short loop with one HIP event per launch boundary, recorded on
    # the launch stream by the engine (unetpp_profile_*); its per-step time is reported next to the un-profiled one
    roofline, agg, prof_ms, executed_gflop = None, {}, None, None
    if not dry:
        psteps = max(2, min(args.steps, 10))
        pdt, _, recs = timed(model, psteps, 1, profile=True)
        prof_ms = pdt / psteps * 1e3
        executed_gflop = sum(r[2] for r in recs) / psteps / B / 1e9        # flops of the launches the engine really made
        for name, ms, fl, by in recs:
            k = name.split("|")[-1]
            a = agg.setdefault(k, [0.0, 0.0, 0.0, 0])
            a[0] += ms; a[1] += fl; a[2] += by; a[3] += 1
        dom = max(agg.items(), key=lambda kv: kv[1][0]) if agg else None
        if dom:
            k, (ms, fl, by, cnt) = dom
            tf = fl / (ms * 1e-3) / 1e12
            # matrix-pipe cycles per product relative to one fp16 MFMA: exact issues 3 MFMAs; exact8 per pair of 16-channel
            # chunks 18 fp16 MFMAs of 32 cycles + 9 scaled fp8 ones of 64 (two taps each; the ninth taps of the two chunks
            # share one) = 1152 / 576
            mfma_mult = {"exact": 3.0, "exact8": 2.0, "fast": 1.0}[args.precision]
            wkey = f"{arch}-c{C}-{H}x{W}-b{B}-{args.precision}"
            traffic, tsrc = measured_traffic(k, wkey)
            roofline = {"bound": "mfma", "kernel": k, "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                        "frac": tf / MFMA_F16_PEAK_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                        "algorithmic_bytes_per_launch": by / cnt,
                        "mfma_issued_tflops": tf * mfma_mult, "mfma_issued_frac": tf * mfma_mult / MFMA_F16_PEAK_TFLOPS,
                        "avg_launch_ms": ms / cnt, "launches": cnt, "profiled_steps": psteps,
                        "algorithmic_gflop_per_launch": fl / cnt / 1e9,
                        "algorithmic_hbm_gbs": by / (ms * 1e-3) / 1e9,
                        "time_share": ms / sum(v[0] for v in agg.values()),
                        "ms_per_step_with_events": prof_ms}

    name = "UNet++" if arch == "nested" else "SimpleUNet"
    out = {
        "metric": "frames/sec 512x512 3-class UNet++ inference; mask vs CPU reference" if (arch, C, H, W) == ("nested", 3, 512, 512)
                  else f"frames/sec {H}x{W} {C}-class {name} inference",
        "value": None if dry else fps, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": DTYPES[args.precision],
        "data": "synthetic" if not dry else "none (--no-engine control-flow rehearsal: nothing was computed)",
        "config": {"workload": f"{name} {C}-class {H}x{W} batch={B}/GPU fp16-MFMA on {world} MI355X, synthetic frames, "
                               f"f32 NCHW in HBM -> uint8 mask in HBM",
                   "arch": arch, "precision": args.precision, "frames_per_gpu": B, "global_batch": B * world,
                   "micro_batch": args.micro_batch or B, "parallelism": f"frame-sharded x{world}, weights replicated (RCCL bcast)"},
        "ms_per_step_samples": [t[0] / args.steps * 1e3 for t in samples],
        # reference-equivalent work (what unetpp.py:104-119 multiplies) and what this build executes (the decoder conv1 of
        # levels 2-3 at low resolution: half their flops), both against the dense fp16 MFMA peak
        "whole_net": {"gflop_per_frame": gflop, "achieved_tflops": fps * gflop / 1e3,
                      "frac_of_f16_mfma_peak": fps * gflop / 1e3 / (MFMA_F16_PEAK_TFLOPS * world),
                      "flops_counted": "reference-equivalent",
                      "executed_gflop_per_frame": executed_gflop,
                      "executed_tflops": None if executed_gflop is None else fps * executed_gflop / 1e3,
                      "executed_frac": None if executed_gflop is None else fps * executed_gflop / 1e3 / (MFMA_F16_PEAK_TFLOPS * world)},
        "roofline": roofline,
        "build": _lib.load().unetpp_version().decode() if not dry else "no-engine",
    }

    # ---- multi-GPU bookkeeping: what really ran where (every rank contributes, rank 0 prints)
    per_rank = gather({"rank": rank, "device": dev_index, "pid": os.getpid(),
                       "frames_per_s": B * args.steps / own_dt, "frames": [lo, hi]})
    if world > 1:
        rates = [p["frames_per_s"] for p in per_rank]
        out["distributed"] = {
            "backend": dist.get_backend(), "rccl_world_size" if backend == "nccl" else "world_size": dist.get_world_size(),
            "launched_by": "bench.py" if os.environ.get("UNETPP_BENCH_LAUNCHED") else "external launcher",
            "devices": [p["device"] for p in per_rank], "pids": [p["pid"] for p in per_rank],
            "frame_shards": [p["frames"] for p in per_rank],
            "per_rank_frames_per_s": {"min": min(rates), "max": max(rates), "all": rates},
            "weight_broadcast": {"bytes": bcast["bytes"], "ms": max_over_ranks(bcast["ms"] or 0.0),
                                 "collectives_in_timed_region": 0},
        }
    if rank == 0 and agg:
        out["kernels"] = {k: {"ms_per_step": v[0] / roofline["profiled_steps"], "tflops": (v[1] / (v[0] * 1e-3) / 1e12) if v[0] else 0.0,
                              "alg_gbs": (v[2] / (v[0] * 1e-3) / 1e9) if v[0] else 0.0,
                              "launches_per_step": v[3] / roofline["profiled_steps"]}
                          for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])}

    ref_cache = {}
    # ---- CPU baseline + parity of the measured mode on the same frames (rank 0, N=1 only)
    if rank == 0 and world == 1 and args.cpu_frames > 0 and not dry:
        base, parity = cpu_baseline(sd, syn, arch, C, H, W, args.cpu_frames, model, torch, ref=ref_cache)
        out["cpu_baseline"] = base
        out["parity"] = parity
    else:
        out["cpu_baseline"] = None

    # ---- informational: the PCIe-inclusive rate (never `value`): pinned uint8 BGR frames -> H2D -> engine (BGR->RGB,
    # /255 fused) -> uint8 masks -> D2H into pinned memory, two engines on two streams so copies overlap compute.
    # With N ranks every rank feeds its own GPU at the same time (what decides >= 6x at 8 GPUs is this host side).
    if not args.no_e2e_leg and not dry:
        try:                                                # informational: a failure here must not cost the headline line
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
            sync(); barrier(); sync()
            t0 = time.perf_counter()
            for i in range(args.steps):
                e2e_step(i)
            sync()
            own_e = time.perf_counter() - t0
            barrier()
            dte = max_over_ranks(time.perf_counter() - t0)
            same = bool(torch.equal(h_out[0], ma.segment(x).cpu()))
            e_rates = gather({"fps": B * args.steps / own_e, "same": same})
            out["end_to_end"] = {"value": B * world * args.steps / dte, "unit": "frames/s", "ms_per_step": dte / args.steps * 1e3,
                                 "path": "pinned uint8 BGR frames (0.79 MB/frame at 512x512) H2D -> engine -> uint8 masks D2H, "
                                         "2 engines on 2 streams per rank; informational, not `value`",
                                 "per_rank_frames_per_s": {"min": min(r["fps"] for r in e_rates), "max": max(r["fps"] for r in e_rates)},
                                 "mask_equals_resident_path": all(r["same"] for r in e_rates)}
            del mb, eng, d_in
        except Exception as exc:                            # noqa: BLE001
            if world > 1:                                   # (the leg holds collectives: a rank that skipped them would hang the others)
                raise
            out["end_to_end"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- informational second leg: the other precision mode on the same workload
    if not args.no_fast_leg and world == 1 and not dry:
        del model
        legs = {}
        for other in [p for p in ("exact", "exact8", "fast") if p != args.precision]:
            try:                                            # a leg is informational: its failure must not cost the headline line
                m2 = make_model(other)
                dt2, _, _ = timed(m2, args.steps, args.warmup, profile=False)
                leg = {"precision": other, "dtype": DTYPES[other], "value": B * args.steps / dt2, "unit": "frames/s",
                       "ms_per_step": dt2 / args.steps * 1e3}
                if rank == 0 and args.cpu_frames > 0:
                    # exact8 is a parity-gated mode (logit_tol 1e-3): its leg is checked on as many frames as the headline
                    _, p2 = cpu_baseline(sd, syn, arch, C, H, W, args.cpu_frames if other == "exact8" else min(args.cpu_frames, 2), m2, torch,
                                         ref=ref_cache)
                    leg["parity"] = p2
                    leg["passes_parity_gate"] = bool(p2["max_abs_logit_err"] < p2["logit_tol"] and p2["flips_outside_near_ties"] == 0)
                del m2
            except Exception as exc:                        # noqa: BLE001
                leg = {"precision": other, "error": f"{type(exc).__name__}: {exc}"}
            legs[other] = leg
        out["legs"] = legs
        out["other_precision"] = legs.get("fast") or legs.get("exact")       # (name kept from earlier rounds)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    return 0


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 16)")
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--classes", type=int, default=None)
    ap.add_argument("--arch", default="nested", choices=["nested", "simple"],
                    help="nested = NestedUNet / UNet++ (the BASELINE metric); simple = SimpleUNet (SURVEY 8(f) row 3)")
    ap.add_argument("--precision", default="exact", choices=["exact", "exact8", "fast"])
    ap.add_argument("--micro-batch", type=int, default=0)
    ap.add_argument("--streams", type=int, default=1)
    ap.add_argument("--cpu-frames", type=int, default=16, help="frames timed on the CPU baseline (0 = skip)")
    ap.add_argument("--no-fast-leg", action="store_true")
    ap.add_argument("--no-e2e-leg", action="store_true", help="skip the PCIe-inclusive informational leg")
    ap.add_argument("--no-engine", action="store_true",
                    help="control-flow rehearsal on CPU (gloo, no GPU, nothing computed, value = null): launcher tests only")
    args = ap.parse_args()
    simple = args.arch == "simple"
    args.batch = args.batch or 16
    args.height = args.height or (256 if simple else 512)
    args.width = args.width or (256 if simple else 512)
    args.classes = args.classes or (7 if simple else 3)
    if args.gpus < 1:
        print("bench.py: --gpus must be >= 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)
    return run_rank(args)


if __name__ == "__main__":
    sys.exit(main())
