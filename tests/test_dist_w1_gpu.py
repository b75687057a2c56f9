"""GPU: the multi-GPU code path of bench.py against RCCL on the ONE device of the test box.  `--gpus 1 --force-dist`
initialises an `nccl` process group of world size 1 on cuda:0 and runs exactly what N > 1 runs: `sharding.distribute_weights` (the device
blob through `dist.broadcast`), both barriers around the timed region, `all_gather_object`, `all_reduce(MAX)`, rank 0's solo anchor of the
same workload and the N > 1 line builder.  What it cannot show is bandwidth between devices — only that no call on the path is wrong for
the backend (device placement of the collectives' tensors, group initialisation with `device_id`, teardown)."""
import json
import os
import subprocess
import sys

import pytest

from tests.helpers import ROOT

pytestmark = pytest.mark.gpu


def test_forced_dist_world_size_one_rccl():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = os.path.join(ROOT, "gpurun_out", "dist_w1_details.json")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--steps", "5", "--warmup", "2", "--details-out", out],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and "nccl" in d["forced_dist"]
    assert "config 4" in d["config"]["workload"] and d["per_rank"][0]["utterances"] == 32
    assert d["value"] > 1000 and d["weight_broadcast_ms"] >= 0                      # B = 32 bf16 on an MI355X: thousands of audio-s/s
    n1 = d["n1_same_workload"]
    assert n1["value"] > 1000 and 0.8 < d["scaling_efficiency"] < 1.25                # the same shard, alone: the same figure
    # the solo anchor and the collective run did the same work
    assert abs(n1["ms_per_step"] - d["ms_per_step"]) < 0.25 * d["ms_per_step"]
