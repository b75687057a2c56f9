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
    # the symbol carries <WM, WN, MI, NI, CK, XR, NLD, NP, RD>: NP = 3 the six-product bf16 form, NP = 2 the two-plane fp16 form ("x3"), RD the ring depth
    cases = {"<4, 1, 1, 2, 32, 128, 0, 3, 2>": "conv1d_x6<128x64>", "<4, 1, 1, 2, 32, 128, 2, 3, 2>": "conv1d_x6<128x64,ld>",
             "<2, 2, 1, 2, 32, 192, 0, 3, 2>": "conv1d_x6<64x128>", "<1, 4, 1, 2, 32, 320, 0, 3, 2>": "conv1d_x6<32x256>",
             "<4, 1, 1, 2, 32, 128, 0, 2, 2>": "conv1d_x3<128x64>", "<4, 1, 1, 2, 32, 128, 2, 2, 2>": "conv1d_x3<128x64,ld>",
             "<4, 1, 1, 2, 32, 128, 4, 2, 4>": "conv1d_x3<128x64,ld4>"}
    for targs, name in cases.items():
        assert family(f"void bv2::conv1d_x6_kernel{targs}(bv2::ConvLaunch, int, int, int)") == name
        assert f'"{name}"' in src, name
        a = [v.strip() for v in targs.strip("<>").split(",")]
        tail = a[6:]                                                                  # trailing defaults (NLD 0, NP 3, RD 2) are left out in the source
        for dflt in ("2", "3", "0"):
            if tail and tail[-1] == dflt and len(tail) == {"2": 3, "3": 2, "0": 1}[dflt]:
                tail = tail[:-1]
        inst = f"launch_x6_variant<{', '.join(a[:6] + tail)}>"
        assert inst in src, inst
    assert family("void bv2::respair_x6_kernel<64, 4, 2>(bv2::FusedLaunch, int)") == "respair_x3<64>"
    assert family("void bv2::respair_x6_kernel<32, 4, 3>(bv2::FusedLaunch, int)") == "respair_x6<32>"


def _strict(txt):
    def fail(name):
        raise AssertionError(f"non-strict JSON constant {name}")
    return json.loads(txt, parse_constant=fail)


def _full_record(n_gpus):
    """The full record of the last committed run, with a per-rank table of n_gpus entries grafted on for N > 1 — the shape
    rank_main() hands to emit()."""
    _, d = _latest_full()
    d = dict(d, n_gpus=n_gpus)
    if n_gpus > 1:
        d["per_rank"] = [dict(rank=r, local_rank=r, device="AMD Instinct MI355X", ms_per_step=18.1234 + r, audio_s_per_step=126.2345,
                              utterances=32, symbols_total=3584 + r, weight_broadcast_ms=123.456,
                              roofline=dict(bound="mfma", kernel="conv_cl_bf16<4x1>", achieved=832.123, peak=2500.0, unit="TFLOP/s",
                                            frac=0.3328, avg_launch_us=575.12)) for r in range(n_gpus)]
        d["ranks_seen"], d["launcher"], d["weight_broadcast_ms"] = n_gpus, "torch.distributed.run", 123.456
        d["n1_same_workload"] = dict(value=7890.12, ms_per_step=16.1234, utterances=32, steps=20)
        d["scaling_efficiency"] = 0.9876
    return d


def _latest_full():
    """Newest committed record that still has the full (pre-headline) shape: launch tables and per-leg roofline blocks."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_details.json")) or
                   [os.path.join(ROOT, "profiles", "r03_k_bench.json")])
    return files[-1], json.load(open(files[-1]))


def test_headline_is_compact_strict_json_for_1_and_8_gpus(tmp_path, capsys):
    """Round 3's stdout line was 36 KB and the driver's record came back `parsed: null`.  The line bench.py prints now goes through
    headline(): < 4 KB, strict JSON, contract keys + scalar roofline / cpu_baseline / parity, per-rank table at N = 8 — and it is
    the LAST line on stdout."""
    import bench
    for n in (1, 8):
        rec = _full_record(n)
        rec["parity"] = dict(rec.get("parity") or {}, wave_max_abs=float("nan"))          # a NaN must not leak into the line
        print("some earlier chatter on stdout")
        txt = bench.emit(rec, str(tmp_path / f"d{n}.json"))
        out = capsys.readouterr().out
        last = out.rstrip("\n").splitlines()[-1]
        assert last == txt and len(last) < (3900 if n == 1 else 4096), len(last)      # 4096 is the limit; the N = 1 line (the one with every block) keeps a margin
        h = _strict(last)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                  "dtype", "data", "config", "roofline", "cpu_baseline"):
            assert k in h, k
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in h["roofline"], k
        for k in ("value", "unit", "cores", "kind", "sample"):
            assert k in h["cpu_baseline"], k
        assert not any(isinstance(v, (list, dict)) for v in h["roofline"].values())          # scalars only: no launch / family tables
        assert "model" not in h["config"] and "workload" in h["config"]
        assert h["parity"].get("wave_max_abs") is None and "NaN" not in last
        if n == 8:
            assert len(h["per_rank"]) == 8 and h["ranks_seen"] == 8 and all(r["roofline"]["frac"] > 0 for r in h["per_rank"])
            assert h["n1_same_workload"]["value"] == 7890.12 and h["scaling_efficiency"] == 0.9876      # the same-workload anchor survives
        else:
            assert h["secondary"]["config3"]["value"] > 0 and "frac" in h["secondary"]["config3"]
            assert abs(h["control_ratio_vs_fp32_mfma"] - rec["value"] / rec["secondary"]["config2_fp32_mfma"]["value"]) < 1e-3
        full = _strict(open(tmp_path / f"d{n}.json").read())                                 # nothing is lost: the full record is on disk
        assert full["roofline"].get("families") == rec["roofline"].get("families")


def test_headline_sheds_blocks_rather_than_overflow():
    import bench
    rec = _full_record(1)
    rec["secondary"] = {f"leg{i}": dict(value=1.0, ms_per_step=2.0, roofline=dict(frac=0.5, bound="mfma", kernel="k" * 40)) for i in range(80)}
    txt = bench.headline(rec)
    assert len(txt) < 4096 and _strict(txt)["secondary"] == "see details" and _strict(txt)["roofline"]["frac"] > 0
