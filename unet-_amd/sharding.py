"""Multi-GPU: frames are independent, so the path shards batch-wise with full weight replicas.
One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  The only collective is
the init-time broadcast of the canonical weight blob from rank 0 over xGMI; there is no steady-state
exchange (optionally an all_gather of uint8 masks when one rank must own every output).
The reference has no distributed code at all; this is an addition, not a port.
"""
from __future__ import annotations

import numpy as np


def shard_range(total: int, rank: int, world: int):
    """Contiguous slice [lo, hi) of `total` frames owned by `rank` (earlier ranks take the remainder)."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_blob(blob_np, nbytes: int, device, src: int = 0):
    """Broadcast the canonical weight blob (uint8) from `src`; returns a uint8 tensor on `device`.
    Ranks other than src pass blob_np=None."""
    import torch
    import torch.distributed as dist
    # RCCL moves device buffers; with the gloo backend (CPU rehearsal) the blob travels through host memory
    comm_dev = device if dist.get_backend() == "nccl" else torch.device("cpu")
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(blob_np)).to(comm_dev)
        assert t.numel() == nbytes
    else:
        t = torch.empty(nbytes, dtype=torch.uint8, device=comm_dev)
    dist.broadcast(t, src=src)
    return t.to(device)


def load_replicated(model, state_dict_or_none, num_classes: int = None, in_channels: int = 3, src: int = 0):
    """Rank `src` folds/packs its state_dict; every rank receives the blob by broadcast and uploads it
    with unetpp_load_weights_device (no host round trip on the receivers).  Works for both architectures
    (NestedUNet and SimpleUNet): the blob size and builder are the model's own."""
    import torch
    import torch.distributed as dist
    from . import _lib
    if num_classes is not None and int(num_classes) != model.num_classes:
        raise ValueError(f"num_classes={num_classes} but the model was built for {model.num_classes}")
    nbytes = int(_lib.load().unetpp_weights_blob_bytes_arch(model._ARCH, model.num_classes, in_channels))
    blob = None
    if dist.get_rank() == src:
        blob = model._check_and_build_blob(state_dict_or_none)
    dev = torch.device(f"cuda:{model._device_index}")
    t = broadcast_blob(blob, nbytes, dev, src)
    model.load_weights_from_device_blob(t)
    return t


def gather_masks(local_mask, world: int):
    """Optional: all_gather of the per-rank uint8 masks (equal shard sizes) -> [world*B_local,H,W]."""
    import torch
    import torch.distributed as dist
    out = [torch.empty_like(local_mask) for _ in range(world)]
    dist.all_gather(out, local_mask)
    return torch.cat(out, 0)
