#!/usr/bin/env python3
"""Average rocprofv3 --pmc counters per launch for kernels matching a substring.
    python tools/pmc_summary.py <dir-with-*_counter_collection.csv> [substr]"""
import collections, csv, glob, sys
def main():
    d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "bv2::"
    for cc in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
        kt = cc.replace("_counter_collection.csv", "_kernel_trace.csv")
        dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); seen = collections.defaultdict(set)
        for r in csv.DictReader(open(cc)):
            kn = r["Kernel_Name"][:70]
            if sub not in kn: continue
            agg[kn][r["Counter_Name"]] += float(r["Counter_Value"]); seen[kn].add(r["Dispatch_Id"])
        for kn, c in agg.items():
            n = len(seen[kn]); du = sum(dur[i] for i in seen[kn]) / n / 1e3
            print(f"{kn}  launches={n} avg_us={du:.2f}")
            for k, v in sorted(c.items()): print(f"    {k:36s} {v / n:14.4e} per launch")
main()
