#!/usr/bin/env python3
"""Per-kernel-family PMC counters of a bench.py run (run ON THE GPU BOX): MFMA utilisation and HBM GB/s as rocprofv3
reports them, next to the HIP-event numbers bench.py prints.

Passes (each `rocprofv3 --kernel-trace --pmc <counters>` only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes — no
sys/hip/hsa tracing next to --pmc):
  1. SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE      MFMA busy cycles (summed over the SIMDs) / kernel cycles
  2. FETCH_SIZE                                    KB read from the fabric (x2 on gfx950, see the guide)
  3. WRITE_SIZE                                    KB written
rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (GUI_ACTIVE / duration = 15-18 "GHz" for the long kernels), so
kernel cycles = GUI_ACTIVE / 8 and the effective clock = that / duration (1.9 GHz under bf16 MFMA load, 2.3 GHz under fp32
MFMA — the DVFS behaviour MI355X_MICROARCH.md describes).
Short launches (< 30 us, or an implied clock above nominal) do not keep all XCDs busy, GUI_ACTIVE/8 under-counts their cycles, and
their mfma_util is computed from the dispatch duration at the nominal 2.4 GHz instead (a lower bound; `mfma_util_basis` says which).
Derived per family:  mfma_util = MFMA_BUSY / (GUI_ACTIVE/8 * 4 SIMDs * 256 CUs)  — cross-check against the HIP-event
numbers: conv1d_mfma<64x64> 0.525 here vs 82 TF / 157.3 TF = 0.52 from bench.py;  hbm_GBps = (2*FETCH + WRITE) / duration,
duration from the same trace (End - Start of the dispatch).

    python tools/collect_pmc.py out.json [--lds] [--traffic traffic.json] [bench.py args...]
`--lds` adds a 4th pass (SQ_LDS_IDX_ACTIVE, SQ_LDS_BANK_CONFLICT): how busy the LDS array is and what conflicts cost.
"""
import collections, csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from collect_traffic import family  # noqa: E402

N_SIMD = 4 * 256
NOMINAL_GHZ = 2.4          # MI355X_MICROARCH.md: peak engine clock
SHORT_US = 30.0            # below this a dispatch does not keep all 8 XCDs busy for its whole life


def one_pass(counters, extra_args):
    d = tempfile.mkdtemp(prefix="bv2pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", d, "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary", "--graph", "0"] + extra_args
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=900)
    tot = collections.defaultdict(collections.Counter)
    n = collections.Counter()
    dur = collections.Counter()
    for cc in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        seen = collections.defaultdict(set)
        rows = list(csv.DictReader(open(cc)))
        inst = collections.Counter((r["Dispatch_Id"], r["Counter_Name"]) for r in rows)   # rows per dispatch = instances
        for r in rows:
            f = family(r["Kernel_Name"])
            if f is None:
                continue
            v = float(r["Counter_Value"])
            if r["Counter_Name"].startswith("GRBM_"):
                v /= inst[(r["Dispatch_Id"], r["Counter_Name"])]
            tot[f][r["Counter_Name"]] += v
            if r["Dispatch_Id"] not in seen[f]:
                seen[f].add(r["Dispatch_Id"])
                if "Start_Timestamp" in r and "End_Timestamp" in r:
                    dur[f] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9
        for f, s in seen.items():
            n[f] += len(s)
    return tot, n, dur


def main():
    out = sys.argv[1]
    extra = sys.argv[2:]
    traffic_out = None
    if "--traffic" in extra:                      # also write the file bench.py's roofline.traffic reads (tools/collect_traffic.py's format)
        i = extra.index("--traffic")
        traffic_out = extra[i + 1]
        extra = extra[:i] + extra[i + 2:]
    lds = "--lds" in extra                        # 4th pass: LDS-array cycles and bank-conflict cycles
    extra = [a for a in extra if a != "--lds"]
    t1, n1, d1 = one_pass(["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"], extra)
    t2, n2, d2 = one_pass(["FETCH_SIZE"], extra)
    t3, n3, d3 = one_pass(["WRITE_SIZE"], extra)
    t4, n4, d4 = one_pass(["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "GRBM_GUI_ACTIVE"],
                          extra) if lds else ({}, {}, {})
    # 5th pass (with --lds): what the waves wait for — wave-cycles, cycles waiting for anything / for an instruction to issue / for LDS
    t5, n5, d5 = one_pass(["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"],
                          extra) if lds else ({}, {}, {})
    res = {"_method": __doc__.split("Derived")[0].strip(), "_bench_args": extra, "kernels": {}}
    for f in sorted(set(t1) | set(t2) | set(t3)):
        row = dict(launches=n1.get(f, 0))
        gui = t1[f].get("GRBM_GUI_ACTIVE", 0.0)
        if gui:
            row["mfma_busy_cycles_per_launch"] = t1[f]["SQ_VALU_MFMA_BUSY_CYCLES"] / max(n1[f], 1)
            row["gui_active_cycles_per_launch"] = gui / max(n1[f], 1)
            util_gui = t1[f]["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8.0 * N_SIMD)
            # GRBM_GUI_ACTIVE / 8 is the kernel's cycle count only while all 8 XCDs are busy for the whole dispatch.  For short launches
            # (a few workgroups, XCDs idle most of the time) it under-counts — "effective clocks" of 3-5 GHz came out of it.
            # Those rows get the duration-based figure instead: busy cycles / (dispatch duration x 2.4 GHz nominal x SIMDs), a LOWER bound
            # of the utilisation (the real clock is <= nominal).
            secs1 = d1[f] / max(n1[f], 1) if d1.get(f) else None
            util_dur = (t1[f]["SQ_VALU_MFMA_BUSY_CYCLES"] / max(n1[f], 1)) / (secs1 * NOMINAL_GHZ * 1e9 * N_SIMD) if secs1 else None
            clock = gui / 8.0 / max(n1[f], 1) / secs1 / 1e9 if secs1 else None
            row["avg_us_in_mfma_pass"] = round(secs1 * 1e6, 2) if secs1 else None
            row["mfma_util_by_duration_at_nominal_clock"] = None if util_dur is None else round(util_dur, 4)
            if secs1 is not None and (secs1 < SHORT_US * 1e-6 or (clock is not None and clock > NOMINAL_GHZ * 1.05)):
                row["mfma_util"] = round(util_dur, 4)
                row["mfma_util_basis"] = f"duration x {NOMINAL_GHZ} GHz (GUI_ACTIVE/8 is not kernel cycles for a launch this short: implied clock {clock:.2f} GHz)"
            else:
                row["mfma_util"] = round(util_gui, 4)
                row["mfma_util_basis"] = "GRBM_GUI_ACTIVE / 8"
        if n2.get(f) and n3.get(f):
            fb = 2 * t2[f]["FETCH_SIZE"] * 1024 / n2[f]
            wb = t3[f]["WRITE_SIZE"] * 1024 / n3[f]
            row["fetch_bytes_per_launch"] = fb
            row["write_bytes_per_launch"] = wb
            secs = (d2[f] / n2[f] + d3[f] / n3[f]) / 2 if d2.get(f) and d3.get(f) else None
            if secs:
                row["avg_us_in_pmc_pass"] = round(secs * 1e6, 2)
                row["hbm_GBps"] = round((fb + wb) / secs / 1e9, 1)
                if gui and row.get("mfma_util_basis") == "GRBM_GUI_ACTIVE / 8":
                    row["effective_clock_GHz"] = round(gui / 8.0 / max(n1[f], 1) / secs / 1e9, 2)
        if lds and f in t4 and t4[f].get("GRBM_GUI_ACTIVE"):
            # SQ_LDS_IDX_ACTIVE = cycles the LDS array worked, SQ_LDS_BANK_CONFLICT = the extra cycles conflicts cost
            # (MI355X_MICROARCH.md, LDS section), both summed over the CUs; kernel cycles = GUI_ACTIVE / 8 as above
            cyc = t4[f]["GRBM_GUI_ACTIVE"] / 8.0
            row["lds_active_frac"] = round(t4[f].get("SQ_LDS_IDX_ACTIVE", 0.0) / (cyc * 256), 4)
            row["lds_bank_conflict_frac_of_lds_cycles"] = round(t4[f].get("SQ_LDS_BANK_CONFLICT", 0.0) / max(t4[f].get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 4)
            row["lds_note"] = "lds_active_frac = SQ_LDS_IDX_ACTIVE / (kernel cycles x 256 CUs): the share of time the LDS array of a CU is busy"
            for cname in ("SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL"):
                if cname in t4[f]:
                    row[cname.lower() + "_frac_of_kernel_cu_cycles"] = round(t4[f][cname] / (cyc * 256), 4)
        if lds and f in t5 and t5[f].get("SQ_WAVE_CYCLES"):
            wc = t5[f]["SQ_WAVE_CYCLES"]
            row["wave_wait_any_frac"] = round(t5[f].get("SQ_WAIT_ANY", 0.0) / wc, 4)
            row["wave_wait_inst_any_frac"] = round(t5[f].get("SQ_WAIT_INST_ANY", 0.0) / wc, 4)
            row["wave_wait_inst_lds_frac"] = round(t5[f].get("SQ_WAIT_INST_LDS", 0.0) / wc, 4)
            row["wave_active_inst_lds_frac"] = round(t5[f].get("SQ_ACTIVE_INST_LDS", 0.0) / wc, 4)
            row["wave_active_inst_vmem_frac"] = round(t5[f].get("SQ_ACTIVE_INST_VMEM", 0.0) / wc, 4)
            row["wave_note"] = "fractions of SQ_WAVE_CYCLES (cycles a wave is resident): waiting on any s_waitcnt / on instruction issue / on LDS issue"
        res["kernels"][f] = row
    json.dump(res, open(out, "w"), indent=1)
    if traffic_out:
        import hashlib
        kdir = os.path.join(ROOT, "bert-vits2_amd", "csrc", "kernels")
        digests = {f: hashlib.sha256(open(os.path.join(kdir, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(kdir)) if f.endswith(".hip")}
        tr = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, the FETCH / WRITE passes of tools/collect_pmc.py) "
                         "around `bench.py --steps 3 --warmup 1`; KB -> bytes; FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md); "
                         "WRITE_SIZE uncalibrated",
              "_bench_args": extra, "source_digests": digests, "kernels": {}}
        for f, row in res["kernels"].items():
            if "fetch_bytes_per_launch" in row:
                tr["kernels"][f] = dict(launches=n2.get(f, 0), fetch_bytes_raw=row["fetch_bytes_per_launch"] / 2, fetch_bytes=row["fetch_bytes_per_launch"],
                                        write_bytes=row["write_bytes_per_launch"], traffic_bytes=row["fetch_bytes_per_launch"] + row["write_bytes_per_launch"])
        json.dump(tr, open(traffic_out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
