#!/usr/bin/env python3
"""Per-launch times of the Generator's five ConvTranspose1d launches (bench.py's upsampling_roofline block: HIP events around each launch in an
eager pass) under different conv_cl_bf16 tile variants, same box (run ON THE GPU BOX):
    python tools/ab_ups.py [--config 3] "<spec>,<cl_generic>,<hc_generic>" ...      ("" = shipped choice; spec pairs separated by ';')"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    cfg = "3"
    if argv and argv[0] == "--config":
        cfg, argv = argv[1], argv[2:]
    for spec in argv:
        with tempfile.NamedTemporaryFile(suffix=".json") as f:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", cfg, "--steps", "12", "--no-secondary", "--no-cpu-baseline", "--details-out", f.name]
            if spec:
                cmd += ["--variants", spec]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(f"{spec or '(shipped)':34s} FAILED rc={r.returncode}", flush=True)
                continue
            d = json.loads(line[-1])
            det = json.load(open(f.name))
        ups = det.get("upsampling_roofline") or {}
        us = [l["us_per_launch"] for l in ups.get("launches", [])]
        names = [l["site"].split("|")[1].split(" ")[0] for l in ups.get("launches", [])]
        print(f"{spec or '(shipped)':34s} {d['ms_per_step']:8.4f} ms/step  ups {sum(us):7.1f} us = " + " + ".join(f"{u:6.1f}" for u in us) + "   " + " ".join(names), flush=True)


if __name__ == "__main__":
    main()
