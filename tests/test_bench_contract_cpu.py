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


def test_pmc_family_names_match_the_names_the_launcher_reports():
    """`bench.py` looks a kernel's PMC traffic up by the variant name `launch_conv1d_x6` reports; `tools/collect_traffic.py` derives the
    same name from the kernel's template arguments in the rocprofv3 trace.  The two must agree (a mismatch silently turns
    `roofline.traffic` into null)."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    from collect_traffic import family
    src = open(os.path.join(root, "bert-vits2_amd", "csrc", "kernels", "conv_x6.hip")).read()
    cases = {"<4, 1, 1, 2, 32, 128, 0>": "conv1d_x6<128x64>", "<4, 1, 1, 2, 32, 128, 2>": "conv1d_x6<128x64,ld>",
             "<2, 2, 1, 2, 32, 192, 0>": "conv1d_x6<64x128>", "<1, 4, 1, 2, 32, 320, 0>": "conv1d_x6<32x256>"}
    for targs, name in cases.items():
        assert family(f"void bv2::conv1d_x6_kernel{targs}(bv2::ConvLaunch, int, int, int)") == name
        assert f'"{name}"' in src, name
        a = [v.strip() for v in targs.strip("<>").split(",")]
        inst = f"launch_x6_variant<{', '.join(a[:6])}" + (f", {a[6]}>" if a[6] != "0" else ">")
        assert inst in src, inst
