#!/usr/bin/env python3
"""Generator time under different conv_x6.hip tiles (bv2_test_set_x6_tuning; run ON THE GPU BOX).
    python tools/tune_x6.py [B] [T_y] ["t256:t128:t64:ck,..."]
Prints ms per Generator pass (event-timed, median of 5 x 10 passes) per setting; tile ids as in bv2_kernels.h TILE_X6_* (0 = shipped)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bert_vits2_amd import hparams as H, lib as L, models, synth  # noqa: E402

NAMES = {0: "shipped", 9: "128x64", 10: "128x64ld", 11: "64x128", 12: "32x256"}   # (256x64, 64x256 and a 32x128 wave tile were measured in round 3 and removed: profiles/r03_tune_x6_*.txt)


def main():
    lib = L.load()
    lib.bv2_test_set_x6_tuning.argtypes = [C.c_int] * 4
    lib.bv2_test_set_x6_tuning.restype = None
    hp = H.default_v23()
    m = models.from_hparams(hp)
    m.load_state_dict(synth.synthetic_state_dict(hp, seed=0, pin_durations=2.5), strict=False)
    m = m.to("cuda").eval()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    Ty = int(sys.argv[2]) if len(sys.argv) > 2 else 384
    z = torch.randn(B, hp.inter_channels, Ty, device="cuda")
    yl = torch.full((B,), Ty, dtype=torch.int64, device="cuda")
    g = torch.randn(B, hp.gin_channels, device="cuda")
    sweep = [(0, 0, 0, 0)]
    sweep += [(t, 0, 0, 0) for t in (9, 11)] + [(0, t, 0, 0) for t in (9, 11)] + [(0, 0, t, 0) for t in (11,)]
    sweep += [(0, 0, 0, 0)]
    if len(sys.argv) > 3:
        sweep = [tuple(int(v) for v in s.split(":")) for s in sys.argv[3].split(",")]
    print(f"B={B} T_y={Ty}")
    for t256, t128, t64, ck in sweep:
        lib.bv2_test_set_x6_tuning(t256, t128, t64, ck)
        for _ in range(3):
            m.stage_generator(z, yl, g)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                m.stage_generator(z, yl, g)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10)
        ts.sort()
        print(f"C256 {NAMES[t256]:9s} C128 {NAMES[t128]:9s} C64 {NAMES[t64]:9s} ck {ck or 'auto':>4}: generator {ts[2]:.4f} ms/pass (min {ts[0]:.4f})", flush=True)
    lib.bv2_test_set_x6_tuning(0, 0, 0, 0)


if __name__ == "__main__":
    main()
