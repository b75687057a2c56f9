"""CPU: the arithmetic claim behind kernels/conv_x6.hip, restated in numpy (no GPU, no library call).

Every fp32 number is the exact sum of three bf16 numbers (round-to-nearest-even at every step); of the nine cross terms of a product
the kernel accumulates six — w1x1, w1x2, w2x1, w1x3, w2x2, w3x1 — each exact in fp32, and drops w2x3 + w3x2 + w3x3.  Checked here:
the dropped part is below 2^-22 of the product for every pair (measured over 4e5 random pairs: worst 2^-24.3 — under half an fp32 ulp —
median 2^-29), and a long dot product accumulated that way in fp32
is as close to the fp64 result as the plain fp32 dot product.  The GPU-side twin is tests/test_x6_gpu.py (the kernel against the
fp32-MFMA kernel); the C-side split is checked bit-exactly in tests/test_cabi_cpu.py."""
import numpy as np


def bf16_rne(v):
    """fp32 array -> the nearest bf16 (ties to even), returned as fp32"""
    u = v.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(v):
    h1 = bf16_rne(v)
    r1 = (v - h1).astype(np.float32)
    h2 = bf16_rne(r1)
    r2 = (r1 - h2).astype(np.float32)
    h3 = bf16_rne(r2)
    return h1, h2, h3


def test_three_bf16_planes_sum_to_the_fp32_value_exactly():
    rng = np.random.default_rng(0)
    v = (rng.standard_normal(200000) * np.exp(rng.uniform(-20, 20, 200000))).astype(np.float32)
    h1, h2, h3 = split3(v)
    assert np.array_equal(h1.astype(np.float64) + h2.astype(np.float64) + h3.astype(np.float64), v.astype(np.float64))
    assert np.all(np.abs(h2) <= np.abs(v) * 2.0 ** -8) and np.all(np.abs(h3) <= np.abs(v) * 2.0 ** -16)


def test_six_cross_terms_are_within_a_fraction_of_an_fp32_ulp_of_the_product():
    rng = np.random.default_rng(1)
    n = 400000
    w = (rng.standard_normal(n) * np.exp(rng.uniform(-8, 8, n))).astype(np.float32)
    x = (rng.standard_normal(n) * np.exp(rng.uniform(-8, 8, n))).astype(np.float32)
    W, X = split3(w), split3(x)
    six = sum(W[i].astype(np.float64) * X[j].astype(np.float64) for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)))
    exact = w.astype(np.float64) * x.astype(np.float64)
    rel = np.abs(six - exact) / np.abs(exact)
    assert rel.max() < 2.0 ** -22, rel.max()
    assert np.median(rel) < 2.0 ** -25
    # every kept term is a bf16 x bf16 product: exact in fp32 (16 significand bits)
    for i, j in ((0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)):
        p64 = W[i].astype(np.float64) * X[j].astype(np.float64)
        ok = np.abs(p64) > 1e-30                                 # (products that underflow fp32 are not the claim)
        assert np.array_equal(p64[ok].astype(np.float32).astype(np.float64), p64[ok])


def test_a_conv_sized_dot_product_is_as_accurate_as_plain_fp32():
    """K = 256 channels x 11 taps, the longest reduction the kernel runs: six-term products accumulated in fp32 in the kernel's order
    (per 16-channel unit: the small terms first) against a plain fp32 multiply-accumulate, both measured against fp64."""
    rng = np.random.default_rng(2)
    K, rows = 256 * 11, 512
    w = (rng.standard_normal((rows, K)) / np.sqrt(K)).astype(np.float32)
    x = (rng.standard_normal((rows, K)) * np.exp(rng.standard_normal((rows, 1)))).astype(np.float32)
    exact = (w.astype(np.float64) * x.astype(np.float64)).sum(1)
    plain = np.zeros(rows, np.float32)
    for k in range(K):
        plain = (plain + w[:, k] * x[:, k]).astype(np.float32)
    W, X = split3(w), split3(x)
    acc = np.zeros(rows, np.float32)
    for u in range(0, K, 16):                                    # one MFMA per (unit, product): 16 exact products summed, then added to fp32
        for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
            part = (W[i][:, u:u + 16].astype(np.float64) * X[j][:, u:u + 16].astype(np.float64)).sum(1)
            acc = (acc + part.astype(np.float32)).astype(np.float32)
    scale = np.abs(exact).max()
    e_plain = np.abs(plain - exact).max() / scale
    e_x6 = np.abs(acc - exact).max() / scale
    assert e_x6 <= 2.0 * e_plain + 1e-7, (e_x6, e_plain)
