"""GPU: kernels/resblock_sw_bf16.hip — a WHOLE ResBlock1 (reference modules.py:296-309) of the C = 64 / C = 32 bf16 Generator stages in one
launch on unpadded, XOR-swizzled LDS rows (an OPTION since it measured slower than the pair kernels: "resblock_sw"), directly against an fp64 restatement with the kernel's rounding points (the same one
tests/test_resblock_c16_gpu.py uses), and at the Generator level against the bf16 oracle with the switch on and off (off = pair by pair,
respair_cl_bf16.hip)."""
import math

import pytest
import torch

from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict, rms
from tests.test_resblock_c16_gpu import _lib, _reference, _run, bf

pytestmark = pytest.mark.gpu

CASES = [
    # C, B, k, dils, L, lens
    (64, 1, 3, (1, 3, 5), 700, None),                 # one tile and a bit (488 output rows per tile at k = 3)
    (64, 2, 7, (1, 3, 5), 1000, [1000, 311]),
    (64, 2, 11, (1, 3, 5), 1300, [1300, 17]),         # four tiles of 392 output rows, one very short utterance
    (64, 1, 11, (1, 3, 5), 393, None),                # one row past a tile boundary
    (64, 1, 5, (2, 1), 333, None),                    # other kernel size / dilation set (C = 64 takes any odd k)
    (32, 1, 3, (1, 3, 5), 1100, None),
    (32, 2, 7, (1, 3, 5), 2100, [2100, 811]),
    (32, 2, 11, (1, 3, 5), 2300, [2300, 5]),
    (32, 1, 11, (1, 3, 5), 905, None),
]


@pytest.mark.parametrize("Cc,B,k,dils,L,lens", CASES)
def test_resblock_sw_kernel_vs_fp64_with_the_same_rounding_points(Cc, B, k, dils, L, lens):
    lib = _lib()
    nd = len(dils)
    g = torch.Generator().manual_seed(1000 * k + L + B + Cc)
    x = bf(torch.randn(B, Cc, L, generator=g))
    w = torch.randn(nd, 2, Cc, Cc, k, generator=g) / math.sqrt(Cc * k)
    b = torch.randn(nd, 2, Cc, generator=g) * 0.3
    lens_t = None if lens is None else torch.tensor(lens, dtype=torch.int64)
    lens_eff = torch.full((B,), L, dtype=torch.int64) if lens_t is None else lens_t
    ref = _reference(x, w, b, k, dils, lens_eff)
    got = _run(lib, 2, x, w, b, k, dils, lens_t)
    valid = (torch.arange(L)[None, :] < lens_eff[:, None])[:, None, :].expand_as(ref)
    assert torch.isfinite(got[valid]).all()
    scale = ref[valid].abs().max().item()
    err = (got - ref)[valid].abs()
    rel = (err.pow(2).mean().sqrt() / ref[valid].pow(2).mean().sqrt()).item()
    frac_1ulp = (err <= 2.0 ** -7 * ref[valid].abs() + 1e-3 * scale).float().mean().item()
    print(f"\n[C={Cc} B={B} k={k} dils={dils} L={L}] rel RMS {rel:.2e}, max |err| {err.max().item():.2e} at scale {scale:.2e}, within 1 ulp {frac_1ulp:.4f}")
    assert rel <= 3e-3 and err.max().item() <= 0.04 * scale and frac_1ulp >= 0.97, (rel, err.max().item(), scale, frac_1ulp)
    if lens is not None:                               # rows past the utterance are never written
        assert torch.isnan(got[~valid]).all()


def test_unsupported_shapes_are_declined():
    import ctypes as C
    lib = _lib()
    P = lambda t: C.c_void_p(t.data_ptr())
    wp = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    for Cc, k, dil in ((32, 5, 1), (128, 3, 1), (64, 11, 7), (16, 3, 1)):      # C = 32 needs k = 3 mod 4; C = 128 / 16 other kernels; 5 * 7 > 32 guard rows
        x = torch.zeros(1, 64, Cc, dtype=torch.bfloat16, device="cuda")
        w = torch.zeros(1, 2, Cc, Cc, k)
        b = torch.zeros(1, 2, Cc)
        da = (C.c_int * 1)(dil)
        assert lib.bv2_test_resblock_cl(None, P(x), P(torch.zeros_like(x)), P(w), P(b), P(wp), 1, Cc, k, da, 1, 64, 0.1, 2, None) == -2


@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged"])
def test_bf16_generator_with_and_without_the_whole_resblock_kernels(name):
    """Stages 2 (C = 64) and 3 (C = 32) of the bf16 Generator through resblock_sw_bf16.hip (default) and pair by pair ("resblock_sw" = 0):
    both within the bf16 bars of the oracle — and bit-identical to each other."""
    from bert_vits2_amd import models
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    ref32 = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                    batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, want_taps=True, **kw)
    with torch.no_grad():
        o16 = O.generator_bf16(sd, hp, ref32["z"] * ref32["y_mask"], ref32["g"], None, {})
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    m.set_generator_dtype(torch.bfloat16)
    outs = {}
    for v in (3, 0, 1, 2):
        m.set_option("resblock_sw", v)
        outs[v] = m.stage_generator(ref32["z"], ref32["y_lengths"], ref32["g"]).cpu()
    m.set_option("resblock_sw", 0)
    sig = rms(o16)
    errs = {v: rms(outs[v] - o16) / sig for v in outs}
    e_pair = rms(outs[3] - outs[0]) / sig
    print(f"\n[{name}] bf16 Generator vs bf16 oracle by resblock_sw = {errs}; whole vs pair-by-pair {e_pair:.3e}")
    assert all(e < 1e-2 for e in errs.values()) and e_pair < 1e-2
    # same unit order (tap-major, groups ascending) and the same rounding points as the pair kernels: bit-identical waveforms
    assert torch.equal(outs[0], outs[3]) and torch.equal(outs[1], outs[3]) and torch.equal(outs[2], outs[3])
