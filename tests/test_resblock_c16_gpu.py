"""GPU: kernels/resblock_c16_bf16.hip — a WHOLE ResBlock1 (reference modules.py:296-309) of the C = 16 Generator stage in one launch on
v_mfma_f32_16x16x32_bf16 (two taps x 16 input channels per instruction), directly against an fp64 restatement with the kernel's rounding
points (every tensor the layer-wise bf16 path stores is rounded to bf16 at the same place), against the 32x32x16 kernel it replaces
(kernels/resblock_cl_bf16.hip, same entry point), and at the Generator level against the bf16 oracle with the switch on and off."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict, rms

pytestmark = pytest.mark.gpu


def _lib():
    from bert_vits2_amd import lib as L
    lib = L.load()
    lib.bv2_test_resblock_cl_pack_bytes.restype = C.c_int64
    lib.bv2_test_resblock_cl_pack_bytes.argtypes = [C.c_int] * 3
    lib.bv2_test_resblock_cl.restype = C.c_int
    lib.bv2_test_resblock_cl.argtypes = [C.c_void_p] * 6 + [C.c_int] * 3 + [C.c_void_p] + [C.c_int] * 2 + [C.c_float, C.c_int, C.c_void_p]
    return lib


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _lrelu(x):
    return torch.where(x < 0, (x.float() * torch.tensor(0.1, dtype=torch.float32)).double(), x)      # fp32 product, like the kernel


def _reference(x, w, b, k, dils, lens):
    """x [B][C][L] (bf16 values, fp64), w [nd][2][C][C][k], b [nd][2][C]; the kernel's rounding points; rows >= lens[b] are zero padding."""
    B, Cc, L = x.shape
    valid = (torch.arange(L)[None, :] < lens[:, None])[:, None, :].double()
    x = x * valid
    for d, dil in enumerate(dils):
        xt = bf(_lrelu(x))
        t = bf(F.conv1d(F.pad(xt, ((k - 1) // 2 * dil,) * 2), bf(w[d, 0]), b[d, 0].double(), dilation=dil))
        h = bf(_lrelu(t)) * valid
        x = bf(F.conv1d(F.pad(h, ((k - 1) // 2,) * 2), bf(w[d, 1]), b[d, 1].double()) + x) * valid
    return x


def _run(lib, variant, x, w, b, k, dils, lens_t):
    B, Cc, L = x.shape
    nd = len(dils)
    xd = x.transpose(1, 2).contiguous().to(torch.bfloat16).cuda()
    out = torch.full((B, L, Cc), float("nan"), dtype=torch.bfloat16, device="cuda")
    wp = torch.empty(lib.bv2_test_resblock_cl_pack_bytes(Cc, k, nd), dtype=torch.uint8, device="cuda")
    da = (C.c_int * nd)(*dils)
    ld = None if lens_t is None else lens_t.cuda()
    rc = lib.bv2_test_resblock_cl(None, P(xd), P(out), P(w.contiguous()), P(b.contiguous()), P(wp), B, Cc, k, da, nd, L, 0.1, variant, P(ld))
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.float().cpu().transpose(1, 2).double()


CASES = [
    # B, k, dils, L, lens
    (1, 3, (1, 3, 5), 700, None),
    (2, 7, (1, 3, 5), 1500, [1500, 811]),
    (2, 11, (1, 3, 5), 2300, [2300, 17]),            # more than two tiles (904 output rows each at k = 11), one very short utterance
    (1, 11, (1, 3, 5), 905, None),                   # one row past a tile boundary
    (3, 5, (2, 1), 333, [333, 1, 200]),              # other kernel size / dilation set / two pairs
    (1, 9, (3,), 64, None),
]


@pytest.mark.parametrize("B,k,dils,L,lens", CASES)
def test_resblock_c16_kernel_vs_fp64_with_the_same_rounding_points(B, k, dils, L, lens):
    lib = _lib()
    Cc, nd = 16, len(dils)
    g = torch.Generator().manual_seed(1000 * k + L + B)
    x = bf(torch.randn(B, Cc, L, generator=g))
    w = torch.randn(nd, 2, Cc, Cc, k, generator=g) / math.sqrt(Cc * k)
    b = torch.randn(nd, 2, Cc, generator=g) * 0.3
    lens_t = None if lens is None else torch.tensor(lens, dtype=torch.int64)
    lens_eff = torch.full((B,), L, dtype=torch.int64) if lens_t is None else lens_t
    ref = _reference(x, w, b, k, dils, lens_eff)
    got = _run(lib, 1, x, w, b, k, dils, lens_t)
    valid = (torch.arange(L)[None, :] < lens_eff[:, None])[:, None, :].expand_as(ref)
    assert torch.isfinite(got[valid]).all()
    scale = ref[valid].abs().max().item()
    err = (got - ref)[valid].abs()
    rel = (err.pow(2).mean().sqrt() / ref[valid].pow(2).mean().sqrt()).item()
    # every stored tensor is bf16: a summation-order difference flips a last bit (2^-8 relative) here and there and the flips propagate
    # through up to six convs; the bulk of the elements must agree to one bf16 ulp
    frac_1ulp = (err <= 2.0 ** -7 * ref[valid].abs() + 1e-3 * scale).float().mean().item()
    print(f"\n[B={B} k={k} dils={dils} L={L}] rel RMS {rel:.2e}, max |err| {err.max().item():.2e} at scale {scale:.2e}, within 1 ulp {frac_1ulp:.4f}")
    assert rel <= 3e-3 and err.max().item() <= 0.04 * scale and frac_1ulp >= 0.97, (rel, err.max().item(), scale, frac_1ulp)
    if lens is not None:                               # rows past the utterance are never written
        assert torch.isnan(got[~valid]).all()
    if k in (3, 7, 11) and dils == (1, 3, 5):          # the 32x32x16 kernel on the same inputs: same rounding points, other summation order
        old = _run(lib, 0, x, w, b, k, dils, lens_t)
        e2 = (old - got)[valid]
        assert (e2.pow(2).mean().sqrt() / ref[valid].pow(2).mean().sqrt()).item() <= 3e-3
        e_old = (old - ref)[valid].abs()
        assert err.pow(2).mean().sqrt().item() <= 1.5 * e_old.pow(2).mean().sqrt().item() + 1e-6 * scale     # not less accurate than its predecessor


def test_unsupported_shapes_are_declined():
    lib = _lib()
    x = torch.zeros(1, 16, 64)
    da = (C.c_int * 1)(7)
    wp = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    w = torch.zeros(1, 2, 16, 16, 11)
    b = torch.zeros(1, 2, 16)
    xd = x.transpose(1, 2).contiguous().to(torch.bfloat16).cuda()
    out = torch.zeros_like(xd)
    # k = 11 with dilation 7 reaches 42 rows outside the tile: more than the 32 guard rows
    assert lib.bv2_test_resblock_cl(None, P(xd), P(out), P(w), P(b), P(wp), 1, 16, 11, da, 1, 64, 0.1, 1, None) == -2
    x32 = torch.zeros(1, 64, 32, dtype=torch.bfloat16, device="cuda")
    w32 = torch.zeros(1, 2, 32, 32, 3)
    b32 = torch.zeros(1, 2, 32)
    da1 = (C.c_int * 1)(1)
    assert lib.bv2_test_resblock_cl(None, P(x32), P(torch.zeros_like(x32)), P(w32), P(b32), P(wp), 1, 32, 3, da1, 1, 64, 0.1, 1, None) == -2


@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged"])
def test_bf16_generator_with_and_without_the_c16_kernel(name):
    """Stage 4 of the bf16 Generator (C = 16) through the new kernel (default) and through the 32x32x16 whole-ResBlock kernel
    ("resblock_c16" = 0): both within the bf16 bars of the oracle (tests/test_bf16_gpu.py), and within a bf16 flip budget of each other."""
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
    for v in (1, 0):
        m.set_option("resblock_c16", v)
        outs[v] = m.stage_generator(ref32["z"], ref32["y_lengths"], ref32["g"]).cpu()
    m.set_option("resblock_c16", 1)
    sig = rms(o16)
    e_new, e_old, e_pair = rms(outs[1] - o16) / sig, rms(outs[0] - o16) / sig, rms(outs[1] - outs[0]) / sig
    print(f"\n[{name}] bf16 Generator vs bf16 oracle: c16 kernel {e_new:.3e}, 32x32x16 kernel {e_old:.3e}; the two against each other {e_pair:.3e}")
    assert e_new < 1e-2 and e_old < 1e-2 and e_pair < 1e-2
    assert not torch.equal(outs[0], outs[1])           # the switch really changed the kernel
