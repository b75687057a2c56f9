"""CPU: the bench line committed under profiles/ (the JSON `python bench.py` printed on the GPU box) carries every key of the driver's
contract, and bench.py's helpers that read the committed PMC evidence do not raise.  No GPU, no timing."""
import glob
import json
import os

from tests.helpers import ROOT


def _latest():
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench.json")))
    assert files, "no committed bench line under profiles/"
    return files[-1], json.load(open(files[-1]))


def test_committed_bench_line_has_the_contract_keys():
    path, d = _latest()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (path, k)
    assert d["unit"] == "audio-seconds/sec" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference")
    # value is consistent with the step time: one 4.458 s utterance per step at N = 1
    assert abs(d["value"] * d["ms_per_step"] * 1e-3 - 196608 / 44100) < 0.02


def test_traffic_lookups_do_not_raise():
    import bench
    for kern, cfg in (("conv1d_mfma<64x64>", 2), ("conv_cl_bf16<4x1>", 3), ("no_such_kernel", 2)):
        t = bench.pmc_traffic(kern, cfg)
        assert isinstance(t, dict) and "bytes_per_launch" in t
    assert bench.bert_traffic() is None or bench.bert_traffic() > 0
