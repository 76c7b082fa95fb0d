"""N>1 path on CPU: world_size-2 gloo run of the weight-blob broadcast and the frame sharding that
bench.py / unet-_amd/sharding.py use on RCCL.  No engine call (no GPU here)."""
import hashlib
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r"""
import hashlib, json, os, sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch, torch.distributed as dist
from unet_amd import packing, sharding, synthetic as syn
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
C = 3
blob = None
nbytes = 32 + 4 * 7846723
if rank == 0:
    sd = syn.make_state_dict(C, 3, True, 2)
    blob = packing.build_blob(sd, C)
    assert blob.nbytes == nbytes
t = sharding.broadcast_blob(blob, nbytes, torch.device("cpu"), src=0)
lo, hi = sharding.shard_range(5, rank, world)          # ragged: 5 frames over 2 ranks
frames = syn.make_frames_u8(hi - lo, 16, 16, "uniform", 99, first=lo)
local = torch.from_numpy(frames[..., 0].copy())
sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(sizes, torch.tensor([hi - lo]))
eq = sharding.shard_range(4, rank, world)
gathered = sharding.gather_masks(torch.full((2, 4, 4), rank, dtype=torch.uint8), world)
print(json.dumps({"rank": rank, "sha": hashlib.sha256(t.numpy().tobytes()).hexdigest(), "range": [lo, hi],
                  "sizes": [int(s) for s in sizes], "frames_sha": hashlib.sha256(frames.tobytes()).hexdigest(),
                  "eq": list(eq), "gathered": gathered[:, 0, 0].tolist()}))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_broadcast_and_shards_world2(tmp_path, syn):
    from unet_amd import packing
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        so, se = p.communicate(timeout=240)
        assert p.returncode == 0, se[-2000:]
        outs.append(__import__("json").loads(so.strip().splitlines()[-1]))
    outs.sort(key=lambda d: d["rank"])
    blob = packing.build_blob(syn.make_state_dict(3, 3, True, 2), 3)
    want = hashlib.sha256(blob.tobytes()).hexdigest()
    assert outs[0]["sha"] == want and outs[1]["sha"] == want           # every rank holds rank 0's weights
    assert outs[0]["range"] == [0, 3] and outs[1]["range"] == [3, 5]   # contiguous, disjoint, complete
    assert outs[0]["sizes"] == [3, 2]
    assert outs[0]["eq"] == [0, 2] and outs[1]["eq"] == [2, 4]
    assert outs[0]["gathered"] == [0, 0, 1, 1]
    all_frames = syn.make_frames_u8(5, 16, 16, "uniform", 99)
    assert outs[0]["frames_sha"] == hashlib.sha256(all_frames[0:3].tobytes()).hexdigest()
    assert outs[1]["frames_sha"] == hashlib.sha256(all_frames[3:5].tobytes()).hexdigest()


def test_shard_range_properties():
    from unet_amd.sharding import shard_range
    for total in (0, 1, 7, 16, 128, 129):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
