"""CPU, world_size 2 over gloo: the N>1 path of bench.py / sharding.py — partitioning, the weight-blob broadcast
(same call that runs over RCCL on GPUs), and the host gather of variable-length audio."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from bert_vits2_amd import sharding


def test_shard_indices_partition_and_balance():
    lengths = [128, 96, 100, 127, 30, 128, 64, 99, 110, 5, 77]
    for world in (1, 2, 4, 8):
        parts = [sharding.shard_indices(lengths, world, r) for r in range(world)]
        assert sorted(i for p in parts for i in p) == list(range(len(lengths)))
        loads = [sum(lengths[i] for i in p) for p in parts]
        if world <= 4:
            assert max(loads) - min(loads) <= max(lengths)
    assert sharding.shard_indices([], 4, 1) == []


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    from bert_vits2_amd import hparams as H, models, synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    hp = H.default_v23(use_transformer_flow=False)
    m = models.from_hparams(hp)
    if rank == 0:
        m.load_state_dict(synth.synthetic_state_dict(hp, 0), strict=False)
    sharding.distribute_weights(m, torch.device("cpu"), src=0)
    blob = m._host_blob
    digest = float(blob[256:].view(torch.float32).double().abs().sum())
    mine = sharding.shard_indices([10, 30, 20, 40, 50], world, rank)
    local = [(i, np.full(i + 1, float(i), dtype=np.float32)) for i in mine]
    allaudio = sharding.gather_audio(local, 5, dst=0)
    q.put((rank, digest, int(blob.numel()), mine, None if allaudio is None else [a.tolist() for a in allaudio]))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_broadcast_and_gather():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, d0, n0, mine0, audio0), (r1, d1, n1, mine1, audio1) = res
    assert n0 == n1 and d0 == d1 and d0 > 0          # rank 1 (zero weights) received rank 0's packed blob bit-exactly
    assert sorted(mine0 + mine1) == [0, 1, 2, 3, 4] and audio1 is None
    assert audio0 == [[float(i)] * (i + 1) for i in range(5)]
