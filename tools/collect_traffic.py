#!/usr/bin/env python3
"""HBM traffic of the hot path's kernels from rocprofv3 PMC counters (run ON THE GPU BOX).

Follows /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 PMC sections): FETCH_SIZE and WRITE_SIZE are collected in
SEPARATE passes (FETCH_SIZE takes 3 of the 4 TCC slots), each pass is `rocprofv3 --kernel-trace --pmc <counter>` only (no
sys/hip/hsa tracing next to --pmc), both counters are reported in KB by rocprofv3, and on gfx950 FETCH_SIZE counts wide
coalesced reads at half their size, so it is DOUBLED before it is compared with a byte count (WRITE_SIZE is uncalibrated
and reported as is).  Output: one JSON mapping bench.py kernel-family names to average bytes per launch.

    python tools/collect_traffic.py profiles/r01_traffic.json        # wraps `python bench.py --steps 3 --warmup 1`
"""
import collections, csv, glob, json, os, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAMILY = {"<2, 2, 2, 2,": "conv1d_mfma<128x128>", "<2, 2, 1, 2,": "conv1d_mfma<64x128>", "<2, 2, 1, 1,": "conv1d_mfma<64x64>",
          "<1, 4, 1, 1,": "conv1d_mfma<32x128>", "<1, 4, 1, 2,": "conv1d_mfma<32x256>"}


def family(kname):
    if "conv1d_mfma_kernel" in kname:
        for k, v in FAMILY.items():
            if "conv1d_mfma_kernel" + k in kname:
                return v
    if "conv1d_x6_kernel<" in kname:            # <WM, WN, MI, NI, CK, XR, NLD, NP> -> the name bench.py / launch_conv1d_x6 report
        a = [int(v) for v in kname.split("conv1d_x6_kernel<")[1].split(">")[0].split(",")]
        form = "x3" if len(a) > 7 and a[7] == 2 else "x6"           # NP = 2: the two-plane fp16 form
        return f"conv1d_{form}<{a[0] * a[2] * 32}x{a[1] * a[3] * 32}{(',ld4' if a[6] == 4 else ',ld') if len(a) > 6 and a[6] > 0 else ''}>"
    if "respair_x6_kernel<" in kname:           # <C, WNT> -> the name bv2_exec.cpp reports
        a = [int(v) for v in kname.split("respair_x6_kernel<")[1].split(">")[0].split(",")]          # <C, WNT, NP>
        return f"respair_{'x3' if len(a) > 2 and a[2] == 2 else 'x6'}<{a[0]}>"
    if "conv1d_splitk_kernel" in kname:
        return "conv1d_splitk<32x32>"
    if "conv_cl_bf16_kernel" in kname:
        for k, v in {"<8, 1, 1, 4,": "conv_cl_bf16<8x1>", "<4, 1, 1, 4,": "conv_cl_bf16<4x1>", "<2, 2, 1, 4,": "conv_cl_bf16<2x2>",
                     "<1, 4, 1, 4,": "conv_cl_bf16<1x4>", "<1, 8, 2, 2,": "conv_cl_bf16<1x8,2x2>", "<4, 2, 2, 2,": "conv_cl_bf16<4x2,2x2>"}.items():
            if "conv_cl_bf16_kernel" + k in kname:
                return v
    if "respair_cl_bf16_kernel<" in kname:      # <WN, WM, NI, G>: C = 16 G -> the name launch_respair_cl_bf16 reports
        a = [int(v) for v in kname.split("respair_cl_bf16_kernel<")[1].split(">")[0].split(",")]
        return f"respair_cl_bf16<{16 * a[3]}>"
    if "respair2_cl_bf16_kernel<" in kname:     # <WN, WM, G>: the 64 x 128 wave-tile form
        a = [int(v) for v in kname.split("respair2_cl_bf16_kernel<")[1].split(">")[0].split(",")[:3]]   # (+ a bool since round 6: the hand-over form)
        return f"respair_cl_bf16<{16 * a[2]},64x128>"
    if "resblock_c16_bf16_kernel" in kname:     # round 5: the C = 16 whole-ResBlock kernel on v_mfma_f32_16x16x32_bf16
        return "resblock_c16_bf16"
    if "resblock_cl_bf16_kernel" in kname:
        return "resblock_cl_bf16<C32>" if "<32," in kname else "resblock_cl_bf16<C16>"
    if "conv_f16_kernel" in kname:
        return "conv_f16"
    if "resblock_fused_kernel" in kname:
        return "resblock_fused"
    if "attention_kernel" in kname:
        return "attention_relpos"
    if "layernorm_kernel" in kname:
        return "layernorm"
    return None


def one_pass(counter, extra_args):
    d = tempfile.mkdtemp(prefix="bv2pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-secondary"] + extra_args
    subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False, timeout=600)
    tot, n = collections.Counter(), collections.Counter()
    for cc in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(cc)):
            if r["Counter_Name"] != counter:
                continue
            f = family(r["Kernel_Name"])
            if f is None:
                continue
            tot[f] += float(r["Counter_Value"])
            seen[f].add(r["Dispatch_Id"])
        for f, s in seen.items():
            n[f] += len(s)
    return {f: (tot[f] / n[f], n[f]) for f in tot if n[f]}


def main():
    out = sys.argv[1]
    extra = sys.argv[2:]
    fetch = one_pass("FETCH_SIZE", extra)
    write = one_pass("WRITE_SIZE", extra)
    import hashlib
    kdir = os.path.join(ROOT, "bert-vits2_amd", "csrc", "kernels")
    digests = {f: hashlib.sha256(open(os.path.join(kdir, f), "rb").read()).hexdigest()[:16] for f in sorted(os.listdir(kdir)) if f.endswith(".hip")}
    res = {"_method": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) around `bench.py --steps 3 --warmup 1`; "
                      "KB -> bytes; FETCH_SIZE x2 (gfx950 wide-read correction, MI355X_MICROARCH.md); WRITE_SIZE uncalibrated",
           "_bench_args": extra,
           "source_digests": digests,      # bench.py only trusts this file for a kernel whose source still has this digest
           "kernels": {}}
    for f in sorted(set(fetch) | set(write)):
        fk, n = fetch.get(f, (0.0, 0))
        wk, _ = write.get(f, (0.0, 0))
        res["kernels"][f] = dict(launches=n, fetch_bytes_raw=fk * 1024, fetch_bytes=2 * fk * 1024, write_bytes=wk * 1024,
                                 traffic_bytes=2 * fk * 1024 + wk * 1024)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
