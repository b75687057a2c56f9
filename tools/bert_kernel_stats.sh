#!/bin/bash
# per-kernel durations of the BERT feature extractor leg (B = 1, S = 53, 22 layers), rocprofv3 --kernel-trace --stats, ON THE GPU BOX
#   bash tools/bert_kernel_stats.sh TAG   ->  gpurun_out/TAG_kernel_stats_bert.txt
TAG=${1:-bert}
REPO=$(cd "$(dirname "$0")/.." && pwd)
cat > /tmp/bert_fw.py <<PY
import sys, torch
sys.path.insert(0, "$REPO")
from bert_vits2_amd.bert_encoder import BertEncoder
from oracle import bert_oracle as BO
cfg = BO.LARGE
enc = BertEncoder(**cfg).load_state_dict(BO.synthetic_state_dict(cfg, 0, layers=22), device="cuda:0")
ids, _ = BO.synthetic_inputs(cfg, [53], 0)
ids = ids.cuda()
for _ in range(30):
    enc(ids)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(50):
    enc(ids)
torch.cuda.synchronize()
print("ms per forward", (time.perf_counter() - t0) / 50 * 1e3)
PY
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_bert
rocprofv3 --kernel-trace --stats -d /tmp/prof_bert --output-format csv -- python /tmp/bert_fw.py > $REPO/gpurun_out/${TAG}_bert_run.log 2>&1
python - <<PY > $REPO/gpurun_out/${TAG}_kernel_stats_bert.txt
import csv, glob
for f in glob.glob("/tmp/prof_bert/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("# 80 forwards in the trace (30 warm-up + 50 timed)")
    for r in rows:
        print(f"{r['Name'][:100]:100s} calls {r['Calls']:>6s} total_us {float(r['TotalDurationNs'])/1e3:10.1f} avg_us {float(r['AverageNs'])/1e3:8.2f} min_us {float(r['MinNs'])/1e3:8.2f} pct {r['Percentage']}")
PY
tail -2 $REPO/gpurun_out/${TAG}_bert_run.log
cat $REPO/gpurun_out/${TAG}_kernel_stats_bert.txt
