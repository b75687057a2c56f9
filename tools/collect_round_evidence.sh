#!/bin/bash
# Round evidence in ONE gpurun call (run ON THE GPU BOX from the repo root):  bash tools/collect_round_evidence.sh r03_e
# PMC passes first (their traffic files are what bench.py's roofline.traffic reads, digest-checked against the kernel sources), then the
# default bench line, then rocprofv3 kernel stats of the same commands.  Everything lands under gpurun_out/ (copied to profiles/ by hand).
set -u
TAG=${1:-r03_e}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
export TMPDIR=/tmp
python tools/collect_pmc.py gpurun_out/${TAG}_pmc_c2.json --lds --traffic profiles/${TAG}_traffic_c2.json > /dev/null 2> gpurun_out/${TAG}_pmc_c2.err
python tools/collect_pmc.py gpurun_out/${TAG}_pmc_c3.json --lds --traffic profiles/${TAG}_traffic_c3.json --config 3 > /dev/null 2> gpurun_out/${TAG}_pmc_c3.err
python tools/collect_traffic_bert.py profiles/${TAG}_traffic_bert.json > /dev/null 2> gpurun_out/${TAG}_traffic_bert.err
cp profiles/${TAG}_traffic_c2.json profiles/${TAG}_traffic_c3.json profiles/${TAG}_traffic_bert.json gpurun_out/ 2>/dev/null
( time python bench.py --details-out gpurun_out/${TAG}_bench_details.json > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err ) 2>&1 | grep real
for C in 2 3; do
  rm -rf /tmp/prof_c$C
  ( cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_c$C -o c$C -- python "$OLDPWD/bench.py" --steps 20 --warmup 5 --no-secondary --no-cpu-baseline $( [ $C = 3 ] && echo "--config 3" ) > /tmp/b$C.json 2> /tmp/b$C.err )
  DB=$(find /tmp/prof_c$C -name "*.db" | head -1)
  python tools/rocprof_summary.py "$DB" 20 > gpurun_out/${TAG}_kernel_stats_c$C.txt
done
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench_details.json"))
print("C2", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "traffic", d["roofline"]["traffic"], "parity", (d.get("parity") or {}).get("wave_rms"))
for k, v in d["secondary"].items():
    if isinstance(v, dict):
        print(k, v.get("value"), v.get("ms_per_step"), v.get("ms_per_request"), v.get("ms_per_sentence"), (v.get("roofline") or {}).get("frac"), (v.get("roofline") or {}).get("traffic"), v.get("error"))
PY
