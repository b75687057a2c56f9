#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace (--stats) rocpd database into the per-kernel table the judge reads.

    python tools/rocprof_summary.py gpurun_out/prof/r1_results.db [steps] > profiles/rNN_<what>.txt
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print(f"# source: {path}")
    print(f"# total kernel time {tot / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches" +
          (f" ({steps} timed steps + warm-up in the trace)" if steps else ""))
    print(f"{'kernel':100s} {'calls':>7s} {'total_us':>12s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>9s} {'pct':>6s} {'vgpr':>5s} {'agpr':>5s} {'lds':>7s}")
    for n, c, s, a, mn, mx, vg, ag, lds in rows:
        print(f"{n[:100]:100s} {c:7d} {s / 1e3:12.1f} {a / 1e3:10.2f} {mn / 1e3:9.2f} {mx / 1e3:9.2f} {100 * s / tot:6.2f} {vg or 0:5d} {ag or 0:5d} {lds or 0:7d}")


if __name__ == "__main__":
    main()
