#!/usr/bin/env python3
"""Per-workgroup timeline of the BERT extractor's GEMM launches (split-K conv kernel, tile id 32032) — run ON THE GPU BOX:
    python tools/timeline_bert.py [prefetch]
One 53-token sentence through the 24 x 1024 model; prints, per GEMM shape, the phase averages (s_memtime ticks) and the launch span."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bert_vits2_amd import bert_synth as BS, lib as L  # noqa: E402
from bert_vits2_amd.bert_encoder import BertEncoder  # noqa: E402


def main():
    lib = L.load()
    lib.bv2_test_conv_timeline.argtypes = [C.c_void_p, C.c_longlong]
    lib.bv2_test_conv_timeline.restype = None
    lib.bv2_test_conv_timeline_report.argtypes = [C.c_void_p, C.c_int]
    lib.bv2_test_conv_timeline_report.restype = C.c_int
    cfg, S, layers = BS.LARGE, 53, 22
    enc = BertEncoder(**cfg).load_state_dict(BS.bert_state_dict(cfg, 0, layers=layers), device="cuda")
    if len(sys.argv) > 1:
        enc.set_option("prefetch", int(sys.argv[1]))
    ids, _ = BS.synthetic_inputs(cfg, [S], 0)
    ids = ids.cuda()
    for _ in range(5):
        enc(ids)
    torch.cuda.synchronize()
    cap = 8 * 1024 * 1024
    buf = torch.zeros(cap, dtype=torch.int64, device="cuda")
    lib.bv2_test_conv_timeline(C.c_void_p(buf.data_ptr()), cap)
    enc(ids)
    torch.cuda.synchronize()
    meta = np.zeros((512, 8), dtype=np.int64)
    n = lib.bv2_test_conv_timeline_report(C.c_void_p(meta.ctypes.data), 512)
    lib.bv2_test_conv_timeline(None, 0)
    raw = buf.cpu().numpy()
    agg = {}
    for i in range(n):
        off, gx, gy, gz, tile, ks, cin, Lc = meta[i]
        s = raw[off: off + 8 * gx * gy * gz].reshape(-1, 8)
        v = s[s[:, 7] == 1]
        if not len(v):
            continue
        xcc = v[:, 5] & 15
        spans = [int(v[xcc == x, 3].max() - v[xcc == x, 0].min()) for x in np.unique(xcc)]
        starts = [int(v[xcc == x, 0].max() - v[xcc == x, 0].min()) for x in np.unique(xcc)]
        key = (int(tile), int(cin), len(v))
        agg.setdefault(key, []).append((np.mean(v[:, 1] - v[:, 0]), np.mean(v[:, 2] - v[:, 1]), np.mean(v[:, 3] - v[:, 2]), np.mean(spans), np.mean(starts)))
    for (tile, cin, wgs), rows in agg.items():
        a = np.array(rows).mean(0)
        print(f"tile {tile} cin {cin:5d} wgs {wgs:4d} launches {len(rows):3d}: prologue {a[0]:7.0f} loop {a[1]:7.0f} epilogue {a[2]:7.0f}  span per XCD {a[3]:8.0f}  start skew {a[4]:7.0f} ticks")


if __name__ == "__main__":
    main()
