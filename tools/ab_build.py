#!/usr/bin/env python3
"""Same-box A/B of two BUILDS of libbv2 (box-to-box spread on the MI355X pool is +-5 %, more than most kernel changes are worth):

  here (no GPU):   python tools/ab_build.py --make <git-rev>       builds bert-vits2_amd/csrc/libbv2_ref.so from <git-rev>'s csrc/
  on the GPU box:  python tools/ab_build.py [--args "<bench args>"] [--rounds N]
                   runs bench.py alternately with the in-tree library and the reference one and prints both series.

The reference library must have the same C ABI and pack layout as the current host code (it is loaded through `bench.py --library`)."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bert-vits2_amd", "csrc")
REF = os.path.join(CSRC, "libbv2_ref.so")


def make(rev):
    sys.path.insert(0, ROOT)
    from bert_vits2_amd import build as B
    tmp = tempfile.mkdtemp(prefix="bv2ref_")
    subprocess.run(f"git -C {ROOT} archive {rev} bert-vits2_amd/csrc include | tar -x -C {tmp}", shell=True, check=True)
    src = os.path.join(tmp, "bert-vits2_amd", "csrc")
    objs = []
    procs = []
    for rel in B.SOURCES:
        if not os.path.exists(os.path.join(src, rel)):
            continue
        obj = os.path.join(tmp, rel.replace("/", "_") + ".o")
        procs.append(subprocess.Popen([B._hipcc()] + B.FLAGS + ["-c", os.path.join(src, rel), "-o", obj]))
        objs.append(obj)
    for p in procs:
        if p.wait():
            raise SystemExit("compile failed")
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", REF] + objs, check=True)
    shutil.rmtree(tmp)
    print("built", REF, "from", rev)


def run(extra, ref):
    det = os.path.join(ROOT, "gpurun_out", f"ab_build_details_{os.getpid()}.json")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-secondary", "--no-cpu-baseline", "--details-out", det] +
                       (["--library", REF] if ref else []) + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line or not os.path.exists(det):
        return None
    d = json.load(open(det))            # the full record (stdout carries the compact headline only)
    os.remove(det)
    fam = {f["name"]: f["ms_per_step"] for f in (d.get("roofline") or {}).get("families", [])}
    return d["ms_per_step"], d["value"], fam


def main():
    if "--make" in sys.argv:
        return make(sys.argv[sys.argv.index("--make") + 1])
    extra, rounds = ["--steps", "30"], 2
    if "--args" in sys.argv:
        extra = sys.argv[sys.argv.index("--args") + 1].split()
    if "--rounds" in sys.argv:
        rounds = int(sys.argv[sys.argv.index("--rounds") + 1])
    for i in range(rounds):
        for ref in (True, False):
            res = run(extra, ref)
            tag = "reference build" if ref else "current build  "
            if res is None:
                print(tag, "FAILED", flush=True)
                continue
            ms, val, fam = res
            print(f"{tag} {ms:9.4f} ms/step {val:9.2f} audio-s/s  " + "  ".join(f"{k.replace('conv_cl_bf16', 'cl').replace('conv1d_mfma', 'mfma')} {v:.3f}" for k, v in fam.items()), flush=True)


if __name__ == "__main__":
    main()
