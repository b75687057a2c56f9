#!/usr/bin/env python3
"""Micro-benchmarks of single kernels through the test-only C entry points (include/bv2_testing.h): HIP-event timing of
N back-to-back launches on one stream.  Used for tuning decisions; numbers are quoted in DESIGN.md / profiles/.

    python tools/kbench.py conv    # conv1d variants on the shapes of the B=1 hot path
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.test_kernels_gpu import _lib, P  # noqa: E402


def time_launch(fn, iters=40, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters        # us per launch


def bench_conv(lib, B, cin, cout, k, dil, L, tile, ksplit=1, lrelu=0.0, res=False):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, cin, L, generator=g).cuda()
    w = torch.randn(cout, cin, k, generator=g) / (cin * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    out = torch.empty(max(ksplit, 1), B, cout, L, device="cuda")
    r = torch.randn(B, cout, L, generator=g).cuda() if res else None
    wp = torch.empty(lib.bv2_test_conv_pack_floats(cin, cout, k), device="cuda")
    slab = B * cout * L
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    call = lambda wh: lib.bv2_test_conv1d(s, P(x), wh, P(bias), P(out), P(wp), B, cin, cout, k, dil, -1, L, tile, lrelu, 0,
                                          P(r), 1 if res else 0, None, None, 0, 0, None, 1, None, None, 1.0, ksplit, slab)
    rc = call(P(w))
    if rc != 0:
        return None
    torch.cuda.synchronize()
    us = time_launch(lambda: call(None))
    flops = 2.0 * B * cin * cout * k * L
    return us, flops / us / 1e6


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "conv"
    lib = _lib()
    if what == "one":              # python tools/kbench.py one cin cout k dil L tile ksplit  (for rocprofv3 --pmc runs)
        cin, cout, k, dil, L, tile, ks = [int(a) for a in sys.argv[2:9]]
        r = bench_conv(lib, 1, cin, cout, k, dil, L, tile, ks, 0.1, tile != 6)
        print("one", sys.argv[2:9], r)
        return
    if what == "conv":
        print("# split-K kernel: waves per workgroup x cross-workgroup ksplit (us per launch, TFLOP/s)")
        for name, (cin, cout, k, L) in dict(ffn1_flow=(192, 768, 5, 384), ffn2_flow=(768, 192, 5, 384), qkv_flow=(192, 594, 1, 384),
                                             ffn1_enc=(192, 768, 3, 128), ffn2_enc=(768, 192, 3, 128), conv_pre=(192, 512, 7, 384)).items():
            for waves in (4, 8, 16):
                lib.bv2_test_set_tuning(waves, 0, 0)
                row = []
                for ks in (1, 2, 4, 8):
                    r = bench_conv(lib, 1, cin, cout, k, 1, L, 6, ks)
                    row.append("   n/a      " if r is None else f"{r[0]:7.2f}us {r[1]:5.1f}TF")
                print(f"{name:10s} waves={waves:2d}  ks=1,2,4,8: " + " | ".join(row))
            lib.bv2_test_set_tuning(0, 0, 0)
            for tile, tn in ((4, "32x128"), (3, "64x64"), (2, "64x128")):
                r = bench_conv(lib, 1, cin, cout, k, 1, L, tile)
                print(f"{name:10s} tiled {tn:8s}: " + ("n/a" if r is None else f"{r[0]:7.2f}us {r[1]:5.1f}TF"))
        print("# LDS-tiled kernel on the Generator's ResBlock shapes (single problem, lrelu + residual)")
        for name, (c, k, dil, L) in dict(s0_k11=(256, 11, 5, 3072), s0_k3=(256, 3, 1, 3072), s1_k11=(128, 11, 5, 24576), s1_k3=(128, 3, 1, 24576),
                                         s2_k11=(64, 11, 5, 49152), s2_k3=(64, 3, 1, 49152), s3_k11=(32, 11, 5, 98304), s3_k3=(32, 3, 1, 98304),
                                         s4_k11=(16, 11, 5, 196608), s4_k3=(16, 3, 1, 196608)).items():
            row = []
            for tile, tn in ((1, "128x128"), (2, "64x128"), (3, "64x64"), (4, "32x128"), (5, "32x256")):
                r = bench_conv(lib, 1, c, c, k, dil, L, tile, 1, 0.1, True)
                row.append(f"{tn}: " + ("   n/a      " if r is None else f"{r[0]:7.2f}us {r[1]:5.1f}TF"))
            print(f"{name:8s} " + " | ".join(row))


if __name__ == "__main__":
    main()
