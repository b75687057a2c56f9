"""CPU: `python bench.py --gpus 2` with NO launcher around it — bench.py spawns its own two ranks, rendezvous on 127.0.0.1 over
gloo, broadcasts the packed weight blob, runs each rank's shard of config 4 (GPU work stubbed at the run_config seam,
tests/bench_seam_cpu.py) and reduces: value = audio of all ranks / the slowest rank's time.  Also the externally-launched form
(torch.distributed.run) through the same code."""
import json
import os
import subprocess
import sys

from tests.helpers import ROOT


def _run(cmd):
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line, printed by rank 0 only
    return json.loads(lines[0])


def _check(d, launcher):
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and d["steps"] == 4
    assert launcher in d["launcher"]
    ranks = sorted(d["per_rank"], key=lambda r: r["rank"])
    assert [r["rank"] for r in ranks] == [0, 1]
    # the fake seam: rank r takes 10*(1+r) ms per step -> the slowest rank (20 ms) sets the time
    assert abs(d["ms_per_step"] - 20.0) < 1e-6 and abs(ranks[0]["ms_per_step"] - 10.0) < 1e-6
    total_audio = sum(r["audio_s_per_step"] for r in ranks)
    assert abs(d["value"] - total_audio / 0.020) < 0.01 * d["value"]
    assert ranks[0]["utterances"] == 32 and ranks[0]["symbols_total"] != ranks[1]["symbols_total"]       # different shards
    assert "config 4" in d["config"]["workload"] and d["config"]["parallelism"] == "utterance-sharded x2"
    assert all(r["roofline"]["frac"] > 0 for r in ranks)                                           # per-GPU roofline
    assert d["weight_broadcast_ms"] >= 0 and d["cpu_baseline"] is None
    # the same-workload single-GPU anchor: rank 0 alone (the fake seam: 8 ms per step) -> efficiency = value / (2 x n1)
    n1 = d["n1_same_workload"]
    assert abs(n1["ms_per_step"] - 8.0) < 1e-6 and abs(n1["value"] - ranks[0]["audio_s_per_step"] / 0.008) < 0.01 * n1["value"]
    assert abs(d["scaling_efficiency"] - d["value"] / (2 * n1["value"])) < 1e-3


def test_self_launched_two_ranks_gloo():
    d = _run([sys.executable, "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--seam", "tests.bench_seam_cpu"])
    _check(d, "self")


def test_torchrun_launched_two_ranks_gloo():
    port = 29600 + os.getpid() % 1500
    d = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "4", "--warmup", "1", "--seam", "tests.bench_seam_cpu"])
    _check(d, "torch.distributed.run")


def test_forced_dist_at_world_size_one_gloo():
    """--gpus 1 --force-dist: the N>1 code path (process group, blob broadcast, barriers, all_reduce(MAX), all_gather_object, the N>1
    line builder) at world size 1 — what tests/test_dist_w1_gpu.py runs against RCCL on the one GPU of the test box."""
    d = _run([sys.executable, "bench.py", "--gpus", "1", "--force-dist", "--steps", "4", "--warmup", "1", "--seam", "tests.bench_seam_cpu"])
    assert d["n_gpus"] == 1 and d["ranks_seen"] == 1 and "config 4" in d["config"]["workload"]
    assert "world size 1" in d["forced_dist"] and d["per_rank"][0]["rank"] == 0
    assert abs(d["ms_per_step"] - 10.0) < 1e-6 and abs(d["n1_same_workload"]["ms_per_step"] - 8.0) < 1e-6
    assert abs(d["scaling_efficiency"] - 0.8) < 1e-3
