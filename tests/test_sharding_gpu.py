"""The N>1 path with the engine in it, on one GPU: every rank receives rank 0's weight blob by broadcast,
uploads it with unetpp_load_weights_device and segments its own shard of the frames; the gathered masks must
equal a single-process run over all frames.  world 2 over gloo (two processes share the card: RCCL refuses two
ranks on one device) and world 1 over nccl (= RCCL: the device-buffer broadcast path)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r"""
import json, os, sys
sys.path.insert(0, sys.argv[1])
backend = sys.argv[2]
import numpy as np, torch, torch.distributed as dist
from unet_amd import sharding, synthetic as syn
from unet_amd.nested_unet import NestedUNet
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda:0")
if backend == "nccl":
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
C, H, W, TOTAL = 3, 64, 96, 5
sd = syn.make_state_dict(C, 3, True, 2) if rank == 0 else None          # only rank 0 has the checkpoint
model = NestedUNet(C, deep_supervision=True, max_batch=3, max_hw=(H, W)).to(dev)
model._ensure_engine(3, H, W)
sharding.load_replicated(model, sd, C)
lo, hi = sharding.shard_range(TOTAL, rank, world)
frames = syn.make_frames_u8(hi - lo, H, W, "smooth", 500, first=lo)
mask = model.segment(torch.from_numpy(frames).to(dev))
torch.cuda.synchronize()
np.save(sys.argv[3] + f"/mask_{rank}.npy", mask.cpu().numpy())
print(json.dumps({"rank": rank, "range": [lo, hi]}))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


@pytest.mark.parametrize("backend,world", [("gloo", 2), ("nccl", 1)])
def test_replicated_weights_and_sharded_frames(backend, world, tmp_path, syn, oracle):
    import torch
    from unet_amd.nested_unet import NestedUNet
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   LOCAL_RANK=str(r), OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, backend, str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    spans = []
    for p in procs:
        so, se = p.communicate(timeout=300)
        assert p.returncode == 0, se[-3000:]
        spans.append(json.loads(so.strip().splitlines()[-1]))
    spans.sort(key=lambda d: d["rank"])
    assert spans[0]["range"][0] == 0 and spans[-1]["range"][1] == 5
    got = np.concatenate([np.load(tmp_path / f"mask_{r}.npy") for r in range(world)])
    # single process, weights loaded the ordinary way
    sd = syn.make_state_dict(3, 3, True, 2)
    m = NestedUNet(3, deep_supervision=True, max_batch=5, max_hw=(64, 96)).to("cuda:0")
    m.load_state_dict(sd, strict=True)
    frames = syn.make_frames_u8(5, 64, 96, "smooth", 500)
    want = m.segment(torch.from_numpy(frames).cuda()).cpu().numpy()
    assert np.array_equal(got, want)                          # same engine arithmetic on every rank: bitwise
    ref = oracle.masks_from_logits(oracle.torch_forward(sd, syn.frames_to_chw_f32(frames)))[0]
    assert (got != ref).sum() <= 2                            # and the CPU reference path (near-tie pixels aside)
