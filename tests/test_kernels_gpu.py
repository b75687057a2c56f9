"""GPU: single HIP kernels (through the test-only C entry points of include/bv2_testing.h) against plain PyTorch fp32
references of the same op, computed on the CPU in fp64 where it matters.  Tolerances are fp32 round-off
(sums of up to a few thousand products): 2e-5 relative to the output scale."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from bert_vits2_amd import lib as L
    lib = L.load()
    f32p = C.c_void_p
    lib.bv2_test_conv_pack_floats.restype = C.c_int64
    lib.bv2_test_conv_pack_floats.argtypes = [C.c_int] * 3
    lib.bv2_test_conv1d.restype = C.c_int
    lib.bv2_test_conv1d.argtypes = ([C.c_void_p] * 6 + [C.c_int] * 8 + [C.c_float, C.c_int, C.c_void_p, C.c_int, C.c_void_p,
                                    C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_float,
                                    C.c_int, C.c_int64])
    lib.bv2_test_resblock_fused.restype = C.c_int
    lib.bv2_test_resblock_fused.argtypes = [C.c_void_p] * 8 + [C.c_int] * 5 + [C.c_float]
    lib.bv2_test_attention.restype = C.c_int
    lib.bv2_test_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3 + [C.c_int] * 5
    lib.bv2_test_layernorm.restype = C.c_int
    lib.bv2_test_layernorm.argtypes = ([C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 3 +
                                       [C.c_int] + [C.c_void_p] * 4 + [C.c_int] * 4 + [C.c_int64])
    lib.bv2_test_spline.restype = C.c_int
    lib.bv2_test_spline.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_float,
                                    C.c_float, C.c_int, C.c_int]
    return lib


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def rel_err(a, b):
    return ((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max().clamp_min(1e-30)).item()


CONV_CASES = [
    # B, cin, cout, k, dil, L, tile
    (1, 32, 32, 3, 1, 200, 4), (1, 32, 32, 3, 1, 200, 5), (2, 64, 64, 7, 3, 300, 2), (2, 64, 64, 7, 3, 300, 3),
    (1, 128, 128, 11, 5, 517, 1), (1, 128, 128, 11, 5, 517, 2), (1, 256, 256, 3, 1, 129, 1), (2, 16, 16, 11, 5, 700, 4),
    (2, 16, 16, 3, 3, 700, 5), (1, 192, 768, 5, 1, 77, 0), (1, 768, 192, 5, 1, 77, 0), (3, 192, 576, 1, 1, 50, 0),
    (2, 1024, 192, 1, 1, 24, 0), (2, 96, 192, 1, 1, 33, 0), (2, 192, 96, 1, 1, 33, 0), (2, 192, 29, 1, 1, 33, 0),
    (2, 256, 1, 1, 1, 40, 0), (1, 192, 512, 7, 1, 64, 0), (1, 192, 384, 5, 1, 31, 0),
    # forced split-K kernel (tile 6) incl. dilation, ragged channel counts, and the LDS-tiled kernel on the same small shapes
    (1, 192, 768, 5, 1, 77, 6), (1, 768, 192, 5, 1, 384, 6), (2, 1024, 192, 1, 1, 24, 6), (2, 96, 192, 1, 1, 33, 6),
    (2, 192, 29, 1, 1, 33, 6), (1, 64, 64, 7, 3, 300, 6), (1, 512, 256, 3, 1, 100, 6), (1, 192, 768, 5, 1, 77, 4),
    (1, 768, 192, 5, 1, 77, 2), (2, 48, 40, 3, 2, 50, 6),
]


@pytest.mark.parametrize("B,cin,cout,k,dil,L,tile", CONV_CASES)
def test_conv1d_mfma_plain(B, cin, cout, k, dil, L, tile):
    lib = _lib()
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + k)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias = torch.randn(cout, generator=g)
    ref = F.conv1d(x.double(), w.double(), bias.double(), padding=(k - 1) // 2 * dil, dilation=dil)
    xd = x.cuda()
    out = torch.full((B, cout, L), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_conv_pack_floats(cin, cout, k), device="cuda")
    rc = lib.bv2_test_conv1d(None, P(xd), P(w), P(bias), P(out), P(wp), B, cin, cout, k, dil, -1, L, tile, 0.0, 0, None, 0,
                             None, None, 0, 0, None, 1, None, None, 1.0, 1, 0)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("ksplit", [2, 4, 8])
@pytest.mark.parametrize("res_mode", [0, 1, 2])
def test_conv1d_splitk_partial_slabs(ksplit, res_mode):
    """K split across workgroups: the result is the SUM of `ksplit` slabs (bias / residual / reverse-subtract ride on
    slab 0, masks are applied to every slab) — what the LayerNorm / embed consumers add up."""
    lib = _lib()
    g = torch.Generator().manual_seed(ksplit * 10 + res_mode)
    B, cin, cout, k, dil, L = 2, 768, 192, 5, 1, 70
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias, res = torch.randn(cout, generator=g), torch.randn(B, cout, L, generator=g)
    mask = (torch.arange(L)[None, :] < torch.tensor([L, L - 19])[:, None]).float()
    y = F.conv1d((x * mask[:, None]).double(), w.double(), bias.double(), padding=(k - 1) // 2) * mask[:, None].double()
    ref = y if res_mode == 0 else (y + res.double() if res_mode == 1 else res.double() - y)
    slab = B * cout * L
    out = torch.full((ksplit, B, cout, L), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_conv_pack_floats(cin, cout, k), device="cuda")
    xd, rd, md = x.cuda(), res.cuda(), mask.cuda()
    rc = lib.bv2_test_conv1d(None, P(xd), P(w), P(bias), P(out), P(wp), B, cin, cout, k, dil, -1, L, 6, 0.0, 0,
                             P(rd) if res_mode else None, res_mode, P(md), P(md), 1, 0, None, 1, None, None, 1.0, ksplit, slab)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_err(out.sum(0), ref) < 2e-5


def test_conv1d_splitk_fused_epilogues():
    """split-K kernel, single slab: lrelu pre-activation with in_scale, per-batch bias, ReLU, masks, residual."""
    lib = _lib()
    g = torch.Generator().manual_seed(77)
    B, cin, cout, k, dil, L = 2, 64, 128, 5, 2, 133
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias, bias2 = torch.randn(cout, generator=g), torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, L, generator=g)
    mask = (torch.arange(L)[None, :] < torch.tensor([L, L - 57])[:, None]).float()
    xin = F.leaky_relu(x.double() * 0.5, 0.1) * mask[:, None].double()
    y = F.conv1d(xin, w.double(), bias.double(), padding=(k - 1) // 2 * dil, dilation=dil) + bias2[:, :, None].double()
    y = torch.relu(y) * mask[:, None].double()
    ref = (y + res.double()) * mask[:, None].double()
    out = torch.full((B, cout, L), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_conv_pack_floats(cin, cout, k), device="cuda")
    xd, rd, md, b2 = x.cuda(), res.cuda(), mask.cuda(), bias2.cuda()
    rc = lib.bv2_test_conv1d(None, P(xd), P(w), P(bias), P(out), P(wp), B, cin, cout, k, dil, -1, L, 6, 0.1, 1,
                             P(rd), 1, P(md), P(md), 1, 1, P(b2), 1, None, None, 0.5, 1, 0)
    assert rc == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5])
def test_conv1d_mfma_fused_epilogues(tile):
    """lrelu pre-activation on a 3-source mean, per-batch bias, ReLU, pre-mask, residual add/rsub, post-mask."""
    lib = _lib()
    g = torch.Generator().manual_seed(tile)
    B, cin, cout, k, dil, L = 2, 64, 128, 5, 2, 333
    xs = [torch.randn(B, cin, L, generator=g) for _ in range(3)]
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias, bias2 = torch.randn(cout, generator=g), torch.randn(B, cout, generator=g)
    res = torch.randn(B, cout, L, generator=g)
    mask = (torch.arange(L)[None, :] < torch.tensor([L, L - 57])[:, None]).float()
    xin = F.leaky_relu((xs[0] + xs[1] + xs[2]).double() * (1.0 / 3.0), 0.1) * mask[:, None].double()
    y = F.conv1d(xin, w.double(), bias.double(), padding=(k - 1) // 2 * dil, dilation=dil) + bias2[:, :, None].double()
    y = torch.relu(y) * mask[:, None].double()
    for res_mode, ref in ((1, (y + res.double()) * mask[:, None].double()), (2, (res.double() - y) * mask[:, None].double())):
        out = torch.full((B, cout, L), float("nan"), device="cuda")
        wp = torch.empty(lib.bv2_test_conv_pack_floats(cin, cout, k), device="cuda")
        xd = [t.cuda() for t in xs]
        rd, md, b2 = res.cuda(), mask.cuda(), bias2.cuda()
        rc = lib.bv2_test_conv1d(None, P(xd[0]), P(w), P(bias), P(out), P(wp), B, cin, cout, k, dil, -1, L, tile, 0.1, 1,
                                 P(rd), res_mode, P(md), P(md), 1, 1, P(b2), 3, P(xd[1]), P(xd[2]), 1.0 / 3.0, 1, 0)
        assert rc == 0
        torch.cuda.synchronize()
        assert rel_err(out, ref) < 2e-5


@pytest.mark.parametrize("B,Cc,k,dil,L", [(1, 32, 11, 5, 1000), (2, 16, 11, 5, 700), (1, 32, 3, 1, 224), (2, 32, 7, 3, 449),
                                           (1, 16, 3, 3, 100), (1, 32, 11, 1, 5000), (1, 8, 5, 2, 300)])
def test_resblock_fused_pair(B, Cc, k, dil, L):
    """out = x + conv2(lrelu(conv1(lrelu(x), dil)), 1): tile seams (L not a multiple of 224), both zero paddings,
    C = 16 (half-empty MFMA rows), C = 8."""
    lib = _lib()
    g = torch.Generator().manual_seed(Cc * 100 + k * 10 + dil)
    x = torch.randn(B, Cc, L, generator=g)
    w1 = torch.randn(Cc, Cc, k, generator=g) / math.sqrt(Cc * k)
    w2 = torch.randn(Cc, Cc, k, generator=g) / math.sqrt(Cc * k)
    b1, b2 = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)
    xd = x.double()
    t = F.conv1d(F.leaky_relu(xd, 0.1), w1.double(), b1.double(), padding=(k - 1) // 2 * dil, dilation=dil)
    ref = xd + F.conv1d(F.leaky_relu(t, 0.1), w2.double(), b2.double(), padding=(k - 1) // 2)
    xg = x.cuda()
    out = torch.full((B, Cc, L), float("nan"), device="cuda")
    wp = torch.empty(2 * lib.bv2_test_conv_pack_floats(Cc, Cc, k), device="cuda")
    assert lib.bv2_test_resblock_fused(None, P(xg), P(out), P(w1), P(b1), P(w2), P(b2), P(wp), B, Cc, k, dil, L, 0.1) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 2e-5


def _ref_attention(qkv, mask, erk, erv, H, W):
    B, C3, T = qkv.shape
    HD = C3 // 3
    D = HD // H
    q, k, v = [t.view(B, H, D, T).transpose(2, 3).double() for t in qkv.split(HD, 1)]
    qs = q / math.sqrt(D)
    s = qs @ k.transpose(-1, -2)
    idx = torch.arange(T)
    rel = idx[None, :] - idx[:, None]
    band = rel.abs() <= W
    ql = qs @ erk.double().t()
    s = s + torch.where(band, ql.gather(-1, (rel + W).clamp(0, 2 * W).expand(B, H, T, T)), torch.zeros((), dtype=torch.float64))
    pair = (mask[:, None, :, None] * mask[:, None, None, :]) != 0
    s = torch.where(pair, s, torch.full((), -1e4, dtype=torch.float64))
    p = torch.softmax(s, -1)
    o = p @ v
    relw = torch.zeros(B, H, T, 2 * W + 1, dtype=torch.float64)
    for r in range(-W, W + 1):
        lo, hi = max(0, -r), min(T, T - r)
        if hi > lo:
            i = torch.arange(lo, hi)
            relw[:, :, lo:hi, r + W] = p[:, :, i, i + r]
    o = o + relw @ erv.double()
    return o.transpose(2, 3).reshape(B, HD, T)


@pytest.mark.parametrize("B,T,lens", [(1, 128, [128]), (2, 100, [100, 37]), (1, 3, [3]), (3, 33, [33, 1, 20]), (1, 400, [400]),
                                      (1, 384, [384]), (2, 250, [250, 129]), (1, 600, [600])])
@pytest.mark.parametrize("f16", [0, 1])
def test_attention_relpos(B, T, lens, f16):
    _attention_case(B, T, lens, f16, 2, 96)


# every head dim the envelope accepts (32 / 64 / 96 / 128) with head counts off the released two (round 6: until then only H = 2, D = 96 / 64 ever
# ran); D = 128 at T > 256 is the four-wave loop form (the eight-wave one spilled), T = 600 the 19-tile loop of every width
@pytest.mark.parametrize("H,D", [(4, 32), (8, 32), (3, 64), (1, 96), (2, 128), (1, 128)])
@pytest.mark.parametrize("B,T,lens", [(2, 100, [100, 37]), (1, 300, [300]), (2, 600, [600, 257]), (1, 5, [5])])
@pytest.mark.parametrize("f16", [0, 1])
def test_attention_relpos_every_head_dim(H, D, B, T, lens, f16):
    _attention_case(B, T, lens, f16, H, D)


def _attention_case(B, T, lens, f16, H, D):
    lib = _lib()
    W = 4
    g = torch.Generator().manual_seed(T + 7 * H + D)
    qkv = torch.randn(B, 3 * H * D, T, generator=g)
    qkv[:, : H * D] *= 3.0                      # sharpen the softmax a little
    erk, erv = torch.randn(2 * W + 1, D, generator=g) * D ** -0.5, torch.randn(2 * W + 1, D, generator=g) * D ** -0.5
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()
    ref = _ref_attention(qkv, mask, erk, erv, H, W)
    out = torch.full((B, H * D, T), float("nan"), device="cuda")
    # kernel input layout: rows [q/sqrt(D) | k | v | per head the 2W+1 relative-key logits], row stride ld (32-aligned),
    # padding columns filled with NaN to prove they are never consumed
    ld = (T + 31) // 32 * 32
    NR = 2 * W + 1
    qs = qkv[:, : H * D] / math.sqrt(D)
    qe = torch.einsum("bhdt,rd->bhrt", qs.view(B, H, D, T), erk).reshape(B, H * NR, T)
    packed = torch.full((B, 3 * H * D + H * NR, ld), float("nan"))
    packed[:, :, :T] = torch.cat([qs, qkv[:, H * D:], qe], 1)
    a = [t.cuda() for t in (packed, mask, erv)]
    fn = lib.bv2_test_attention_f16 if f16 else lib.bv2_test_attention
    assert fn(None, P(a[0]), ld, P(a[1]), P(a[2]), P(out), B, H, D, T, W) == 0
    torch.cuda.synchronize()
    valid = mask[:, None, :].bool().expand_as(ref)          # padded query rows are masked downstream
    # fp16 mode: q, k, v and the probabilities are rounded to fp16 (2^-11 relative each) before the two matrix products
    tol = 4e-3 if f16 else 2e-5
    assert ((out.cpu().double() - ref).abs()[valid].max() / ref.abs().max()).item() < tol
    assert torch.isfinite(out).all()


@pytest.mark.parametrize("C,T,B", [(192, 77, 2), (256, 40, 1), (192, 5, 3)])
def test_layernorm_family(C, T, B):
    lib = _lib()
    g = torch.Generator().manual_seed(C + T)
    a, add, res = (torch.randn(B, C, T, generator=g) for _ in range(3))
    gamma, beta, vec = torch.randn(C, generator=g), torch.randn(C, generator=g), torch.randn(B, C, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor([T, max(1, T - 3), 1][:B])[:, None]).float()

    def ln(v):
        return F.layer_norm(v.double().transpose(1, 2), (C,), gamma.double(), beta.double(), 1e-5).transpose(1, 2)

    dev = lambda t: t.cuda()
    # mode 0: LN(a+add) + vec, masked
    ref = (ln(a + add) + vec[:, :, None].double()) * mask[:, None].double()
    out = torch.empty(B, C, T, device="cuda")
    t = [dev(x) for x in (a, add, gamma, beta, vec, mask)]
    assert lib.bv2_test_layernorm(None, P(t[0]), P(t[1]), 0, None, None, 1, None, P(t[2]), P(t[3]), 0, None, P(t[4]), P(t[5]),
                                  P(out), B, C, T, 1, 0) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-5
    # mode 0 fed by 4 partial slabs whose sum is `a`
    parts = torch.randn(4, B, C, T, generator=g)
    parts[0] = a - parts[1] - parts[2] - parts[3]
    pd_ = parts.cuda()
    assert lib.bv2_test_layernorm(None, P(pd_), P(t[1]), 0, None, None, 1, None, P(t[2]), P(t[3]), 0, None, P(t[4]), P(t[5]),
                                  P(out), B, C, T, 4, B * C * T) == 0
    torch.cuda.synchronize()
    assert rel_err(out, ref) < 1e-5
    # mode 1: gelu(LN(depthwise k3 dil conv(a*mask))) ; then res + .
    for dil in (1, 3, 9):
        dww, dwb = torch.randn(C, 1, 3, generator=g), torch.randn(C, generator=g)
        y = F.conv1d((a * mask[:, None]).double(), dww.double(), dwb.double(), padding=dil, dilation=dil, groups=C)
        ref = res.double() + F.gelu(ln(y))
        t2 = [dev(x) for x in (dww, dwb, res)]
        assert lib.bv2_test_layernorm(None, P(t[0]), None, 1, P(t2[0]), P(t2[1]), dil, P(t[5]), P(t[2]), P(t[3]), 1, P(t2[2]),
                                      None, None, P(out), B, C, T, 1, 0) == 0
        torch.cuda.synchronize()
        assert rel_err(out, ref) < 1e-5


def test_spline_inverse_kernel():
    from oracle import bv2_oracle as O
    lib = _lib()
    g = torch.Generator().manual_seed(5)
    B, T, K = 2, 300, 10
    params = torch.randn(B, 32, T, generator=g) * 3
    z = torch.randn(B, 2, T, generator=g) * 4           # some beyond the +-5 tails
    mask = (torch.arange(T)[None, :] < torch.tensor([T, T - 40])[:, None]).float()
    h = params[:, :29].transpose(1, 2)
    ref1 = O.rq_spline_inverse(z[:, 1].double(), (h[..., :K] / math.sqrt(192)).double(), (h[..., K:2 * K] / math.sqrt(192)).double(),
                               h[..., 2 * K:].double()) * mask.double()
    zd, pd, md = z.cuda(), params.cuda(), mask.cuda()
    assert lib.bv2_test_spline(None, P(zd), 0, 1, P(pd), 32, P(md), math.sqrt(192), 5.0, B, T) == 0
    torch.cuda.synchronize()
    assert (zd[:, 1].cpu().double() - ref1).abs().max().item() < 2e-5
    assert torch.equal(zd[:, 0].cpu(), z[:, 0] * mask)


@pytest.mark.parametrize("cin,cout,k,dil,L,ksplit", [(192, 768, 5, 1, 384, 1), (768, 192, 5, 1, 384, 8), (192, 768, 3, 1, 128, 1),
                                                     (768, 192, 3, 1, 77, 4), (64, 64, 7, 3, 300, 1), (48, 40, 3, 2, 50, 1),
                                                     (192, 512, 7, 1, 384, 1), (256, 256, 3, 1, 128, 2)])
def test_splitk_lds_staged_form_is_bit_identical_to_the_register_form(cin, cout, k, dil, L, ksplit):
    """The split-K kernel's LDSX form (X tile staged once in LDS, all taps read it back) runs the same MFMA sequence on the same
    operand values as the form that re-loads X per tap from global memory: results must be bitwise equal (and right)."""
    lib = _lib()
    lib.bv2_test_set_tuning.argtypes = [C.c_int, C.c_int, C.c_long]
    lib.bv2_test_set_tuning.restype = None
    B = 2
    g = torch.Generator().manual_seed(cin + cout + k)
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias = torch.randn(cout, generator=g)
    mask = (torch.arange(L)[None, :] < torch.tensor([L, max(1, L - 9)])[:, None]).float()
    xd, md = x.cuda(), mask.cuda()
    wp = torch.empty(lib.bv2_test_conv_pack_floats(cin, cout, k), device="cuda")
    outs = []
    for tune in (0, 100):
        lib.bv2_test_set_tuning(tune, 0, 0)
        slab = B * cout * L
        out = torch.full((ksplit, B, cout, L), float("nan"), device="cuda")
        rc = lib.bv2_test_conv1d(None, P(xd), P(w), P(bias), P(out), P(wp), B, cin, cout, k, dil, -1, L, 6, 0.1, 0, None, 0,
                                 P(md), None, 0, 0, None, 1, None, None, 1.0, ksplit, slab)
        assert rc == 0
        torch.cuda.synchronize()
        outs.append(out.sum(0))
    lib.bv2_test_set_tuning(0, 0, 0)
    assert torch.equal(outs[0], outs[1])
    ref = F.conv1d(F.leaky_relu(x.double(), 0.1) * mask.double()[:, None], w.double(), bias.double(), padding=(k - 1) // 2 * dil, dilation=dil)
    assert rel_err(outs[0], ref) < 2e-5
