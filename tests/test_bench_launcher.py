"""`python bench.py --gpus N` must start its own N ranks (BASELINE config 3 is 8 x 16 frames; the driver's N>1 command
may or may not come with a launcher).  CPU rehearsal: gloo backend, `--no-engine` worker (no GPU, nothing computed,
value = null) — what is tested is the launcher: fresh children with RANK/WORLD_SIZE/MASTER_*, one JSON line relayed
from rank 0, per-rank bookkeeping gathered, exit codes propagated."""
import json
import os
import subprocess
import sys

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def run(args, extra_env=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env.update(extra_env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=env)


def test_self_launch_two_ranks_gloo():
    r = run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-engine", "--batch", "4", "--height", "32", "--width", "32"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{")      # stdout = the one JSON line of rank 0, nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["value"] is None and "no-engine" in d["data"]   # a rehearsal never reports a rate
    dd = d["distributed"]
    assert dd["world_size"] == 2 and dd["backend"] == "gloo" and dd["launched_by"] == "bench.py"
    assert len(set(dd["pids"])) == 2 and os.getpid() not in dd["pids"]          # two fresh processes
    assert dd["frame_shards"] == [[0, 4], [4, 8]]            # contiguous shards of the global batch
    assert dd["weight_broadcast"]["collectives_in_timed_region"] == 0
    assert d["config"]["global_batch"] == 8


def test_world_size_mismatch_and_missing_gpus_fail_loudly():
    # an external launcher that disagrees with --gpus
    r = run(["--gpus", "2", "--no-engine"], {"RANK": "0", "WORLD_SIZE": "3", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert r.returncode == 2 and "WORLD_SIZE=3" in r.stderr
    # more ranks than HIP devices (none here): refused before any rank starts
    import torch
    if torch.cuda.device_count() < 2:
        r = run(["--gpus", "2", "--steps", "1"])
        assert r.returncode == 2 and "needs 2 HIP devices" in r.stderr and not r.stdout.strip()


def test_failing_rank_ends_the_job():
    # rank 1 cannot join (its WORLD view is broken by the injected failure): the launcher stops the others and reports
    r = run(["--gpus", "2", "--steps", "2", "--no-engine", "--batch", "2", "--height", "16", "--width", "16"],
            {"UNETPP_BENCH_FAIL_RANK": "1", "UNETPP_BENCH_TIMEOUT": "120"})
    assert r.returncode != 0
    assert "rank exit codes" in r.stderr
