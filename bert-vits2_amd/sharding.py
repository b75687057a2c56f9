"""Multi-GPU: utterance-level sharding, one process per GPU, RCCL only for the one-time weight broadcast.

``infer()`` couples nothing across batch elements (every op is per utterance; masks are per utterance, reference
commons.py:119-123), so the path shards naturally over utterances: each rank synthesises its own
slice, the variable-length audio is gathered on the host, and the steady state has NO collective.  The reference has
no multi-GPU inference at all (its only communication is DDP/NCCL in train_ms.py:71-78, 250-258).
"""
from __future__ import annotations

import time
from typing import List, Optional, Sequence

import torch


def _dist():
    import torch.distributed as dist
    return dist if (dist.is_available() and dist.is_initialized()) else None


def shard_indices(lengths: Sequence[int], world: int, rank: int) -> List[int]:
    """Length-balanced assignment of utterances to ranks: longest first, dealt in a serpentine so every rank gets
    about the same number of symbols (the idea of the reference's DistributedBucketSampler, data_utils.py:305-335,
    applied to inference).  Deterministic; the union over ranks is a partition of range(len(lengths))."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    mine = []
    for pos, idx in enumerate(order):
        rnd, k = divmod(pos, world)
        owner = k if rnd % 2 == 0 else world - 1 - k
        if owner == rank:
            mine.append(idx)
    return sorted(mine)


def distribute_weights(model, device: torch.device, src: int = 0) -> float:
    """Make ``model`` runnable on ``device`` on every rank; returns the seconds spent in the broadcast.

    world == 1: pack + upload.  world > 1: only ``src`` needs real parameters — it folds/packs them once and the packed
    blob (one contiguous buffer, ~270 MB fp32) is broadcast with RCCL (backend "nccl"); the other ranks just attach it.
    xGMI is point-to-point, so a ring broadcast is bound by one ~153 GB/s link: a few ms, once per model load.
    """
    dist = _dist()
    world = dist.get_world_size() if dist else 1
    rank = dist.get_rank() if dist else 0
    lib = model._ensure_handle()
    nbytes = int(lib.bv2_packed_bytes(model._handle))
    if device.type != "cuda":
        # CPU (gloo) path used by the tests: the blob is still broadcast, but cannot be attached without a GPU
        blob = model.pack_host_blob() if rank == src else torch.empty(nbytes, dtype=torch.uint8)
        t0 = time.perf_counter()
        if dist is not None:
            dist.broadcast(blob, src=src)
        model._host_blob = blob
        return time.perf_counter() - t0
    with torch.cuda.device(device):
        if rank == src:
            blob = model.pack_host_blob().to(device)
        else:
            blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
        dt = 0.0
        if dist is not None:                 # also at world size 1 when a process group exists: the same RCCL call, a self-copy
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            dist.broadcast(blob, src=src)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        model.attach_blob(blob)
    return dt


def gather_audio(local: List, n_total: int, dst: int = 0) -> Optional[List]:
    """Host-side gather of ``[(utterance_index, 1-D float32 numpy audio), ...]`` from every rank to ``dst``; returns the
    list ordered by utterance index on ``dst`` (None elsewhere).  Audio is variable length, so this is an object
    gather over the host, not a device collective."""
    dist = _dist()
    if dist is None:
        out = [None] * n_total
        for i, a in local:
            out[i] = a
        return out
    bucket = [None] * dist.get_world_size() if dist.get_rank() == dst else None
    dist.gather_object(local, bucket, dst=dst)
    if dist.get_rank() != dst:
        return None
    out = [None] * n_total
    for part in bucket:
        for i, a in part:
            out[i] = a
    return out
