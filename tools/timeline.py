#!/usr/bin/env python3
"""Per-workgroup timeline of the fp32 conv kernels inside a real Generator / flow pass (run ON THE GPU BOX).

    python tools/timeline.py gpurun_out/timeline_c2.npz [--batch 1 --symbols 128]

Every conv launch of one ``infer()`` records, per workgroup, s_memtime at kernel start / after the prologue / after the main loop /
at the end plus HW_ID and XCC_ID (include/bv2_testing.h bv2_test_conv_timeline).  The raw stamps are saved as .npz (analysed
offline with tools/timeline_report.py — analysis needs no GPU); a short per-launch summary is printed."""
import argparse
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bert_vits2_amd import hparams as H, lib as L, models, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--symbols", type=int, default=128)
    ap.add_argument("--bf16", action="store_true", help="bf16 Generator + fp16 flow (configs 3 / 5)")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=V", help="bv2_set_option before the pass")
    a = ap.parse_args()
    lib = L.load()
    lib.bv2_test_conv_timeline.argtypes = [C.c_void_p, C.c_longlong]
    lib.bv2_test_conv_timeline.restype = None
    lib.bv2_test_conv_timeline_report.argtypes = [C.c_void_p, C.c_int]
    lib.bv2_test_conv_timeline_report.restype = C.c_int
    hp = H.default_v23()
    m = models.from_hparams(hp)
    m.load_state_dict(synth.synthetic_state_dict(hp, seed=0, pin_durations=2.5), strict=False)
    m = m.to("cuda").eval()
    if a.bf16:
        m.set_generator_dtype(torch.bfloat16)
        m.set_flow_dtype(torch.float16)
    for kv in a.option:
        key, val = kv.split("=")
        m.set_option(key, int(val))
    b = {k: v.cuda() for k, v in synth.synthetic_batch([a.symbols] * a.batch).items()}
    kw = dict(noise_scale=0.6, noise_scale_w=0.9, sdp_ratio=0.0, length_scale=1.0)
    call = lambda: m.infer(b["x"], b["x_lengths"], b["sid"], b["tone"], b["language"], b["bert"], b["ja_bert"], b["en_bert"], **kw)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    cap = 48 * 1024 * 1024
    buf = torch.zeros(cap, dtype=torch.int64, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.bv2_test_conv_timeline(C.c_void_p(buf.data_ptr()), cap)
    e0.record()
    call()
    e1.record()
    torch.cuda.synchronize()
    meta = np.zeros((512, 8), dtype=np.int64)
    n = lib.bv2_test_conv_timeline_report(C.c_void_p(meta.ctypes.data), 512)
    lib.bv2_test_conv_timeline(None, 0)
    meta = meta[:n]
    raw = buf.cpu().numpy()
    used = int(meta[-1, 0] + 8 * meta[-1, 1] * meta[-1, 2] * meta[-1, 3]) if n else 0
    np.savez_compressed(a.out, meta=meta, raw=raw[:used], step_ms=np.array([e0.elapsed_time(e1)]))
    print(f"{n} conv launches, {used // 8} workgroup slots, instrumented step {e0.elapsed_time(e1):.3f} ms")
    for i in range(n):
        off, gx, gy, gz, tile, ks, cin, Lc = meta[i]
        s = raw[off: off + 8 * gx * gy * gz].reshape(-1, 8)
        v = s[s[:, 7] == 1]
        if not len(v):
            continue
        span = v[:, 3].max() - v[:, 0].min()
        if tile <= -90000:
            # respair_cl_bf16.hip (tile id -(90000 + C)): slot 6 = taps | (ticks of conv1's GEMM) << 16; "loop" = conv1 + h epilogue + conv2
            for kk in np.unique(v[:, 6] & 0xffff):
                w = v[(v[:, 6] & 0xffff) == kk]
                c1 = w[:, 6] >> 16
                print(f"     respair C={-tile - 90000} k={kk:2d}: {len(w):6d} wgs  stage {np.mean(w[:, 1] - w[:, 0]):7.0f}  conv1 {np.mean(c1):7.0f}  "
                      f"h-epilogue + conv2 {np.mean(w[:, 2] - w[:, 1] - c1):7.0f}  epilogue {np.mean(w[:, 3] - w[:, 2]):7.0f}  "
                      f"life {np.mean(w[:, 3] - w[:, 0]):8.0f} ticks (MFMA-only time of the two GEMMs: {2 * kk * (-tile - 90000) // 16 * 4 * 32} cycles)")
        print(f"#{i:3d} tile {tile:6d} k={ks & 255},{(ks >> 8) & 255},{(ks >> 16) & 255} cin {cin:4d} L {Lc:6d} wgs {len(v):5d}/{len(s):5d} span {span:8d} ticks "
              f"prologue {np.mean(v[:, 1] - v[:, 0]):8.0f} loop {np.mean(v[:, 2] - v[:, 1]):9.0f} epilogue {np.mean(v[:, 3] - v[:, 2]):8.0f}")


if __name__ == "__main__":
    main()
