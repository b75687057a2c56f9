#!/usr/bin/env python3
"""Offline analysis of tools/timeline.py output (no GPU needed).

    python tools/timeline_report.py gpurun_out/timeline_c2.npz [launch indices...]

s_memtime is per-XCD (the eight counters have different offsets), so spans are taken per XCD.  For each launch: workgroups per CU,
per-problem (k) phase times, MFMA-pipe utilisation = sum of MFMA cycles issued by all waves / (span x SIMDs), and how the span splits
into 'all CUs busy' vs tail.  conv_x6 launches (tile id 6xxxxxx): phases per k, ticks per 12-MFMA unit, time at chunk switches, and how
many workgroups actually share a CU while it is busy."""
import sys

import numpy as np


def hw(v):
    hwid, xcc = v[:, 4], v[:, 5] & 15
    simd = (hwid >> 4) & 3
    cu = (hwid >> 8) & 15
    sh = (hwid >> 12) & 1
    se = (hwid >> 13) & 7
    return xcc, se, sh, cu, simd


def main():
    z = np.load(sys.argv[1])
    meta, raw = z["meta"], z["raw"]
    sel = [int(a) for a in sys.argv[2:]] or range(len(meta))
    for i in sel:
        off, gx, gy, gz, tile, ks, cin, Lc = meta[i]
        s = raw[off: off + 8 * gx * gy * gz].reshape(-1, 8)
        v = s[s[:, 7] == 1]
        if not len(v):
            continue
        xcc, se, sh, cu, simd = hw(v)
        cukey = ((xcc * 8 + se) * 2 + sh) * 16 + cu
        ncu = len(np.unique(cukey))
        spans = []
        for x in np.unique(xcc):
            m = xcc == x
            spans.append(v[m, 3].max() - v[m, 0].min())
        span = float(np.mean(spans))
        bm, bn = tile // 1000, tile % 1000
        kk = [int((ks >> (8 * j)) & 255) for j in range(3) if (ks >> (8 * j)) & 255]
        line = f"#{i:3d} {bm}x{bn} k={kk} cin {cin} L {Lc}: {len(v)} wgs on {ncu} CUs ({len(v) / ncu:.2f}/CU), span {span:9.0f} (per-XCD {min(spans)}..{max(spans)})"
        if tile // 1000000 == 6:
            # conv_x6.hip: tile id 6000000 + BM*1000 + BN; slot 6 = taps | (ticks spent at chunk switches / barriers) << 16
            t = tile - 6000000
            kx, sw = v[:, 6] & 0xffff, v[:, 6] >> 16
            conc, busy = [], []
            for kcu in np.unique(cukey):
                mm = cukey == kcu
                a, b = v[mm, 0].astype(np.int64), v[mm, 3].astype(np.int64)
                o = np.argsort(a)
                a, b = a[o], b[o]
                cs, ce, uni = a[0], b[0], 0
                for s0, e0 in zip(a[1:], b[1:]):
                    if s0 <= ce:
                        ce = max(ce, e0)
                    else:
                        uni += ce - cs
                        cs, ce = s0, e0
                uni += ce - cs
                conc.append((b - a).sum() / uni)
                busy.append(b.max() - a.min())
            life = v[:, 3] - v[:, 0]
            print(f"#{i:3d} conv_x6 {t // 1000}x{t % 1000} k={kk} cin {cin} L {Lc}: {len(v)} wgs on {ncu} CUs ({len(v) / ncu:.2f}/CU), workgroups sharing a CU while it is busy "
                  f"{np.mean(conc):.2f}, CU busy span {np.mean(busy):8.0f} (max {np.max(busy)}), longest workgroup {life.max()}, mean {life.mean():8.0f}")
            for k in kk:
                m = kx == k
                if m.any():
                    units = (cin / 16) * k
                    lo = v[m, 2] - v[m, 1]
                    print(f"      k{k}: n {m.sum()} prologue {np.mean(v[m, 1] - v[m, 0]):7.0f} loop {lo.mean():8.0f} ({lo.mean() / units:6.1f} per 12-MFMA unit; chunk switches / barrier waits "
                          f"{np.mean(sw[m]):7.0f}) epilogue {np.mean(v[m, 3] - v[m, 2]):7.0f} life {np.mean(life[m]):8.0f}")
            continue
        if 99000 <= tile < 100000:
            t = tile - 99000
            wn, ni, io = t // 100, (t // 10) % 10, t % 10
            k = int(ks & 255)
            cyc = (cin / 16) * k * ni * 32
            print(f"#{i:3d} fp16 conv {wn}w NI{ni} in_ct={io >> 1} out_ct={io & 1} k={k} cin {cin} L {Lc}: {len(v)} wgs on {ncu} CUs ({len(v) / ncu:.1f}/CU): first stage {np.mean(v[:, 1] - v[:, 0]):7.0f} "
                  f"rest (gemm + later chunks) {np.mean(v[:, 2] - v[:, 1]):8.0f} (MFMA-only {cyc:7.0f}) epilogue {np.mean(v[:, 3] - v[:, 2]):7.0f} life {np.mean(v[:, 3] - v[:, 0]):8.0f}")
            continue
        if 77000 <= tile < 78000 or 88000 <= tile < 89000:
            what = "attention" if tile < 78000 else "resblock_fused"
            print(f"#{i:3d} {what} id {tile} k={kk} D/C {cin} L {Lc}: {len(v)} wgs on {ncu} CUs ({len(v) / ncu:.2f}/CU): phase1 {np.mean(v[:, 1] - v[:, 0]):7.0f} "
                  f"phase2 {np.mean(v[:, 2] - v[:, 1]):8.0f} phase3 {np.mean(v[:, 3] - v[:, 2]):7.0f} life {np.mean(v[:, 3] - v[:, 0]):8.0f}")
            if what == "attention":
                print(f"      merge: wait+stats {np.mean(v[:, 5] - v[:, 2]):7.0f}  slot rounds {np.mean(v[:, 6] - v[:, 5]):7.0f}  normalise+store {np.mean(v[:, 3] - v[:, 6]):7.0f}")
            if what == "resblock_fused":
                for k in kk:
                    m = v[:, 6] == k
                    if m.any():
                        print(f"      k{k}: n {m.sum()} stage {np.mean(v[m, 1] - v[m, 0]):7.0f} conv1 {np.mean(v[m, 2] - v[m, 1]):8.0f} conv2+store {np.mean(v[m, 3] - v[m, 2]):8.0f}")
            continue
        if tile < 0:
            # bf16 channels-last conv: tile id -(WN*1000 + WM*100 + MI*10 + NI); MFMA cycles per wave = (cin/16)*k units * MI*NI * 32
            t = -tile
            wn, wm, mi, ni = t // 1000, (t // 100) % 10, (t // 10) % 10, t % 10
            print(f"#{i:3d} bf16 {wn}x{wm} MI{mi} NI{ni} k={kk} cin {cin} L {Lc}: {len(v)} wgs on {ncu} CUs ({len(v) / ncu:.1f}/CU)")
            tot = 0.0
            for k in kk:
                m = v[:, 6] == k
                if not m.any():
                    continue
                cyc = (cin / 16) * k * mi * ni * 32
                tot += m.sum() * cyc * (wn * wm / 4.0)
                lo = v[m, 2] - v[m, 1]
                print(f"      k{k}: n {m.sum()} stage {np.mean(v[m, 1] - v[m, 0]):7.0f} gemm {lo.mean():8.0f} (MFMA-only {cyc:7.0f}, x{lo.mean() / cyc:4.2f}) "
                      f"epilogue {np.mean(v[m, 3] - v[m, 2]):7.0f}  life {np.mean(v[m, 3] - v[m, 0]):8.0f}")
            continue
        if tile != 32032:
            # MFMA cycles per wave of a workgroup of problem p: (cin/8 groups) * k taps * 4 MFMAs * (MI*NI) * 64 cycles; waves = 4
            mi_ni = (bm // 64) * (bn // 64) if bm >= 64 and bn >= 64 else (1 if bm * bn <= 32 * 128 else 2)
            tot = 0.0
            parts = []
            for p, k in enumerate(kk):
                m = v[:, 6] == k
                if not m.any():
                    continue
                cyc = (cin / 8) * k * 4 * mi_ni * 64
                tot += m.sum() * cyc
                lo = v[m, 2] - v[m, 1]
                parts.append(f"k{k}: n {m.sum()} prol {np.mean(v[m, 1] - v[m, 0]):6.0f} loop {lo.mean():8.0f} (ideal alone {cyc:7.0f}, x{lo.mean() / cyc:4.2f}) epi {np.mean(v[m, 3] - v[m, 2]):6.0f}")
            util = tot / (span * ncu)            # each CU has 4 SIMDs and each workgroup puts one wave on each: cycles per SIMD
            line += f"  MFMA util {util:5.3f}"
            print(line)
            for p in parts:
                print("      " + p)
            # concurrency profile on the busiest XCD: how many workgroups are in their main loop over time
            x = np.bincount(xcc).argmax()
            m = xcc == x
            t0 = v[m, 0].min()
            ev = np.concatenate([np.stack([v[m, 1] - t0, np.ones(m.sum())], 1), np.stack([v[m, 2] - t0, -np.ones(m.sum())], 1)])
            ev = ev[np.argsort(ev[:, 0])]
            conc = np.cumsum(ev[:, 1])
            dt = np.diff(ev[:, 0], append=ev[-1, 0])
            ncu_x = len(np.unique(cukey[m]))
            tot_t = ev[-1, 0] - ev[0, 0]
            for thr in (1.0, 2.0, 3.0):
                frac = dt[conc >= thr * ncu_x].sum() / tot_t
                print(f"      XCD {x}: fraction of loop span with >= {thr:.0f} workgroups/CU in the main loop: {frac:5.2f}")
        else:
            U = v[:, 6]
            print(line)
            print(f"      units/wave {U.mean():5.1f}  prol {np.mean(v[:, 1] - v[:, 0]):6.0f}  loop {np.mean(v[:, 2] - v[:, 1]):7.0f} (ideal {U.mean() * 256:6.0f})  epi {np.mean(v[:, 3] - v[:, 2]):6.0f}"
                  f"  wg life {np.mean(v[:, 3] - v[:, 0]):7.0f}")


if __name__ == "__main__":
    main()
