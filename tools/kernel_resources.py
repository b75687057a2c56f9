#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the HIP sources, from hipcc's -Rpass-analysis=kernel-resource-usage remarks (no GPU
needed):    python tools/kernel_resources.py [file.hip ...]        (default: every kernels/*.hip)

Why it exists (round 3): a kernel that spills even two registers gets a scratch segment, and a dispatch with a scratch segment drains
the queue on this stack — one such kernel (6 launches per step) cost +0.3 ms per 4.6 ms step while the sum of kernel times fell.
`--check` exits non-zero if any kernel the product launches by default (see ALLOWED_SCRATCH for the known exceptions) has ScratchSize > 0;
tests/test_cabi_cpu.py runs it."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KDIR = os.path.join(ROOT, "bert-vits2_amd", "csrc", "kernels")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "--cuda-device-only", "-c", "-Rpass-analysis=kernel-resource-usage"]
# variants that are only reachable through tuning hooks / shapes no accepted model has (16-wave split-K, the NI = 2 bf16 tile): their
# spills cannot reach the hot path.  (Round 6: head dim 128 left this list — its long-sequence form runs on four waves and does not spill.)
ALLOWED_SCRATCH = ("conv1d_splitk_kernelILb1ELi16E", "conv1d_splitk_kernelILb0ELi16E",
                   "conv_cl_bf16_kernelILi4ELi1ELi1ELi2E")


def table(path):
    r = subprocess.run(["hipcc"] + FLAGS + [path, "-o", os.devnull], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    rows, cur = [], None
    for line in r.stdout.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = dict(name=m.group(1))
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return rows, r.returncode


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    check = "--check" in sys.argv
    files = args or sorted(glob.glob(os.path.join(KDIR, "*.hip")))
    bad = []
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(8, len(files))) as ex:      # hipcc runs are independent processes
        results = list(ex.map(table, files))
    for f, (rows, rc) in zip(files, results):
        if rc:
            print(f"{f}: hipcc failed")
            bad.append(f)
            continue
        for k in rows:
            short = re.sub(r"^_ZN3bv2\d+", "", k["name"])
            scratch = k.get("ScratchSize", 0)
            flag = ""
            if scratch:
                allowed = any(a in k["name"] for a in ALLOWED_SCRATCH)
                flag = "  <-- scratch (not on the default path)" if allowed else "  <-- SCRATCH"
                if not allowed:
                    bad.append(k["name"])
            if not check or scratch:
                print(f"{os.path.basename(f):22s} {short[:70]:70s} vgpr {k.get('VGPRs', 0):3d} agpr {k.get('AGPRs', 0):3d} sgpr {k.get('TotalSGPRs', 0):3d} "
                      f"scratch {scratch:4d} occ {k.get('Occupancy', 0)} lds {k.get('LDS Size', 0)}{flag}")
    if check and bad:
        print("kernels with a scratch segment on the default path:", *bad, sep="\n  ")
        sys.exit(1)


if __name__ == "__main__":
    main()
