#!/usr/bin/env python3
"""Generator time at BASELINE config 2 (B=1, T_y=384) under different tile-count targets of the fp32 conv's auto picker
(bv2_test_set_tuning; run ON THE GPU BOX).  Prints ms per Generator pass (event-timed, median of 5 x 10 passes)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bert_vits2_amd import hparams as H, lib as L, models, synth  # noqa: E402


def main():
    lib = L.load()
    lib.bv2_test_set_tuning.argtypes = [C.c_int, C.c_int, C.c_long]
    lib.bv2_test_set_tuning.restype = None
    hp = H.default_v23()
    m = models.from_hparams(hp)
    m.load_state_dict(synth.synthetic_state_dict(hp, seed=0, pin_durations=2.5), strict=False)
    m = m.to("cuda").eval()
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    Ty = 384
    z = torch.randn(B, hp.inter_channels, Ty, device="cuda")
    yl = torch.full((B,), Ty, dtype=torch.int64, device="cuda")
    g = torch.randn(B, hp.gin_channels, device="cuda")
    names = {0: "default tile picker, cost-balanced placement", -7: "128x64 tile where C_out % 128 == 0", -8: "128x64 tile at C = 128 only",
             -9: "64x128 tile at C = 64", -10: "64x128 tile at C = 128", -11: "128x128 tile at C = 128"}
    # argv[2]: comma list of "force_ck:tile_target" pairs to run instead of the default sweep (force_ck 16 = 16-channel X chunks:
    # half the LDS per workgroup, twice the barriers; +100 = plain z-major placement); the environment variable BV2_CONV_FLAGS=1
    # turns the epilogue-operand prefetch of the 64x64 tile off (same-box A/B: run this script twice in one gpurun call)
    sweep = ((0, 0), (100, 0), (0, -7), (0, -8), (0, -9), (0, -10), (0, -11))
    if len(sys.argv) > 2:
        sweep = tuple(tuple(int(v) for v in pair.split(":")) for pair in sys.argv[2].split(","))
    print(f"BV2_CONV_FLAGS={os.environ.get('BV2_CONV_FLAGS', '0')}")
    for occ, target in sweep:
        lib.bv2_test_set_tuning(0, occ, target)
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
        label = 'plain z-major placement (no snake)' if occ == 100 else names[target]
        if occ % 100 == 16:
            label += " + 16-channel X chunks"
        print(f"{label}: generator {ts[2]:.4f} ms/pass (min {ts[0]:.4f})", flush=True)
    lib.bv2_test_set_tuning(0, 0, 0)


if __name__ == "__main__":
    main()
