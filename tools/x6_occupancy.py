#!/usr/bin/env python3
"""Workgroups per CU the HIP runtime grants the shipped kernels/conv_x6.hip variants (hipOccupancyMaxActiveBlocksPerMultiprocessor with
each variant's block size and dynamic LDS; run ON THE GPU BOX).  Measured in round 3: 128x64 3, 128x64 with loader waves 2, 64x128 2,
32x256 2 — what a launch actually reaches is another matter (tools/timeline.py: the 6-wave loader workgroups share a CU 1.14-fold).

    python tools/x6_occupancy.py
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from bert_vits2_amd import lib as L  # noqa: E402


def main():
    lib = L.load()
    torch.zeros(1, device="cuda")                      # a context on device 0
    a = (C.c_int * 4)()
    lib.bv2_test_x6_occupancy.restype = None
    lib.bv2_test_x6_occupancy.argtypes = [C.POINTER(C.c_int)]
    lib.bv2_test_x6_occupancy(a)
    print("conv_x6 workgroups per CU: 128x64", a[0], " 128x64 + loader waves", a[1], " 64x128", a[2], " 32x256", a[3])


if __name__ == "__main__":
    main()
