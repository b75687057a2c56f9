#!/usr/bin/env python3
"""Same-box A/B of bv2_set_option switches through bench.py (run ON THE GPU BOX, inside ONE gpurun call so both sides see the same
GPU / clocks):   python tools/ab.py [--steps N] [--args "<extra bench args>"] "key=val[,key=val]" "key=val" ...
Each positional argument is one run (an empty string = shipped defaults); runs are repeated in the given order, so list the baseline
first and last to see the drift.  Prints audio-s/s and ms/step per run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    steps, extra = "40", []
    while argv and argv[0].startswith("--"):
        if argv[0] == "--steps":
            steps = argv[1]
        elif argv[0] == "--args":
            extra = argv[1].split()
        argv = argv[2:]
    for spec in argv:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", steps, "--no-secondary", "--no-cpu-baseline"] + extra
        for kv in filter(None, spec.split(",")):
            cmd += ["--option", kv]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"{spec or '(defaults)':40s} FAILED rc={r.returncode}", flush=True)
            continue
        d = json.loads(line[-1])
        rf = d.get("roofline") or {}
        print(f"{spec or '(defaults)':40s} {d['value']:9.2f} audio-s/s  {d['ms_per_step']:8.4f} ms/step  dominant {rf.get('kernel')} frac {rf.get('frac')}", flush=True)


if __name__ == "__main__":
    main()
