"""GPU: the round-4 fused kernels DIRECTLY against fp64 restatements of the reference code they replace (until now they were
held to the layer-wise kernels, which were held to the oracle — the chain made the layer-wise fallbacks load-bearing for parity):

* kernels/respair_cl_bf16.hip — one (dilated conv, conv) pair of ResBlock1 with its residual (reference modules.py:296-309), bf16
  channels-last, both forms, against fp64 with the kernel's rounding points;
* kernels/respair_x6.hip — the same pair in fp32 on the bf16 matrix core from exact three-way splits, against fp64: its error must not
  exceed twice that of an ordinary fp32 evaluation of the same expression (the claim tests/test_x6_gpu.py makes for conv_x6.hip);
* kernels/flow_boundary.hip — LayerNorm-2 + `post` with the reverse update of x1 + the next coupling's `pre` (reference
  attentions.py:118-120, models.py:121-132), against fp64."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from bert_vits2_amd import lib as L
    lib = L.load()
    lib.bv2_test_respair_cl_pack_bytes.restype = C.c_int64
    lib.bv2_test_respair_cl_pack_bytes.argtypes = [C.c_int] * 2
    lib.bv2_test_respair_cl.restype = C.c_int
    lib.bv2_test_respair_cl.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_int, C.c_void_p]
    lib.bv2_test_respair_x6_pack_bytes.restype = C.c_int64
    lib.bv2_test_respair_x6_pack_bytes.argtypes = [C.c_int] * 2
    lib.bv2_test_respair_x6.restype = C.c_int
    lib.bv2_test_respair_x6.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_void_p]
    lib.bv2_test_respair_x3.restype = C.c_int
    lib.bv2_test_respair_x3.argtypes = [C.c_void_p] * 6 + [C.c_int] * 5 + [C.c_float, C.c_void_p]
    lib.bv2_test_flow_boundary_pack_floats.restype = C.c_int64
    lib.bv2_test_flow_boundary_pack_floats.argtypes = [C.c_int]
    lib.bv2_test_flow_boundary.restype = C.c_int
    lib.bv2_test_flow_boundary.argtypes = ([C.c_void_p, C.c_void_p, C.c_int, C.c_int64] + [C.c_void_p] * 4 + [C.c_int64] + [C.c_void_p] * 6 +
                                           [C.c_int] * 3)
    return lib


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def bf(t):
    return t.to(torch.bfloat16).to(torch.float64)


def _lrelu32(x):                                      # the kernels form x * 0.1 in fp32
    return torch.where(x < 0, (x.float() * torch.tensor(0.1, dtype=torch.float32)).double(), x)


@pytest.mark.parametrize("C_,k,dil,L,B,form,lens", [
    (32, 11, 5, 1300, 2, 0, [1300, 601]),
    (32, 3, 1, 515, 1, 0, None),
    (64, 7, 3, 700, 2, 1, [700, 33]),
    (64, 11, 1, 300, 1, 0, None),                    # form 0 at C = 64
    (128, 11, 5, 520, 2, 1, None),
    (128, 3, 3, 257, 1, 1, None),
    (256, 7, 1, 300, 1, 1, None),
])
def test_respair_cl_bf16_vs_fp64_with_the_same_rounding_points(C_, k, dil, L, B, form, lens):
    lib = _lib()
    g = torch.Generator().manual_seed(C_ * 17 + k * 3 + dil + L)
    x = bf(torch.randn(B, C_, L, generator=g))
    w = torch.randn(2, C_, C_, k, generator=g) / math.sqrt(C_ * k)
    b = torch.randn(2, C_, generator=g) * 0.3
    lens_t = None if lens is None else torch.tensor(lens, dtype=torch.int64)
    le = torch.full((B,), L, dtype=torch.int64) if lens_t is None else lens_t
    valid = (torch.arange(L)[None, :] < le[:, None])[:, None, :].double()
    xin = x * valid
    t = bf(F.conv1d(F.pad(bf(_lrelu32(xin)), ((k - 1) // 2 * dil,) * 2), bf(w[0]), b[0].double(), dilation=dil))
    h = bf(_lrelu32(t)) * valid
    ref = bf(F.conv1d(F.pad(h, ((k - 1) // 2,) * 2), bf(w[1]), b[1].double()) + xin)
    xd = x.transpose(1, 2).contiguous().to(torch.bfloat16).cuda()
    out = torch.full((B, L, C_), float("nan"), dtype=torch.bfloat16, device="cuda")
    wp = torch.empty(lib.bv2_test_respair_cl_pack_bytes(C_, k), dtype=torch.uint8, device="cuda")
    ld = None if lens_t is None else lens_t.cuda()
    rc = lib.bv2_test_respair_cl(None, P(xd), P(out), P(w.contiguous()), P(b.contiguous()), P(wp), B, C_, k, dil, L, 0.1, form, P(ld))
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = out.float().cpu().transpose(1, 2).double()
    vm = valid.bool().expand_as(ref)
    assert torch.isfinite(got[vm]).all()
    err = (got - ref)[vm].abs()
    scale = ref[vm].abs().max().item()
    rel = (err.pow(2).mean().sqrt() / ref[vm].pow(2).mean().sqrt()).item()
    ulp1 = (err <= 2.0 ** -7 * ref[vm].abs() + 1e-3 * scale).float().mean().item()
    print(f"\n[C={C_} k={k} d={dil} L={L} form={form}] rel RMS {rel:.2e}, max |err| {err.max().item():.2e} at scale {scale:.2e}, within 1 ulp {ulp1:.4f}")
    assert rel <= 2e-3 and err.max().item() <= 0.04 * scale and ulp1 >= 0.98, (rel, err.max().item(), scale, ulp1)


@pytest.mark.parametrize("C_,k,dil,L,B,lens", [
    (16, 11, 5, 900, 2, [900, 411]),
    (16, 3, 1, 300, 1, None),
    (32, 7, 3, 700, 2, None),
    (64, 11, 1, 515, 1, None),
    (64, 3, 5, 260, 2, [260, 7]),
    (128, 7, 1, 200, 1, None),
])
@pytest.mark.parametrize("form", ["x6", "x3"])          # six bf16 products / three scaled-fp16 products (bv2_kernels.h): the same bars
def test_respair_x6_vs_fp64(C_, k, dil, L, B, lens, form):
    lib = _lib()
    g = torch.Generator().manual_seed(C_ * 31 + k * 5 + dil + L)
    # operands spanning e^+-3 in scale per channel: a split-bf16 kernel that dropped a plane would show here
    x = torch.randn(B, C_, L, generator=g) * torch.exp(torch.randn(1, C_, 1, generator=g) * 1.5)
    w = torch.randn(2, C_, C_, k, generator=g) / math.sqrt(C_ * k) * torch.exp(torch.randn(2, C_, 1, 1, generator=g))
    b = torch.randn(2, C_, generator=g) * 0.3
    lens_t = None if lens is None else torch.tensor(lens, dtype=torch.int64)
    le = torch.full((B,), L, dtype=torch.int64) if lens_t is None else lens_t
    valid = (torch.arange(L)[None, :] < le[:, None])[:, None, :]

    def pair(dt):
        v = valid.to(dt)
        xi = x.to(dt) * v
        lr = lambda t: torch.where(t < 0, t * torch.tensor(0.1, dtype=torch.float32).to(dt), t)
        t = F.conv1d(F.pad(lr(xi), ((k - 1) // 2 * dil,) * 2), w[0].to(dt), b[0].to(dt), dilation=dil)
        h = lr(t) * v
        return F.conv1d(F.pad(h, ((k - 1) // 2,) * 2), w[1].to(dt), b[1].to(dt)) + xi

    ref, ref32 = pair(torch.float64), pair(torch.float32).double()
    out = torch.full((B, C_, L), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_respair_x6_pack_bytes(C_, k), dtype=torch.uint8, device="cuda")
    ld = None if lens_t is None else lens_t.cuda()
    xdev = x.cuda()
    fn = lib.bv2_test_respair_x6 if form == "x6" else lib.bv2_test_respair_x3
    rc = fn(None, P(xdev), P(out), P(w.contiguous()), P(b.contiguous()), P(wp), B, C_, k, dil, L, 0.1, P(ld))
    assert rc == 0, rc
    torch.cuda.synchronize()
    got = out.cpu().double()
    vm = valid.expand_as(ref)
    assert torch.isfinite(got[vm]).all()
    # per output row (channel): the kernel's max error against fp64 vs the fp32 evaluation's
    zero = torch.zeros_like(ref)
    e6 = torch.where(vm, (got - ref).abs(), zero).amax(dim=(0, 2))          # (rows past the utterance are never written: NaN there)
    e32 = torch.where(vm, (ref32 - ref).abs(), zero).amax(dim=(0, 2))
    rowscale = torch.where(vm, ref.abs(), zero).amax(dim=(0, 2))
    print(f"\n[C={C_} k={k} d={dil} L={L}] {form} pair max err / row scale {float((e6 / rowscale).max()):.2e}, fp32 evaluation {float((e32 / rowscale).max()):.2e}")
    # 2x the fp32 evaluation's error + a few fp32 ulps of the row's scale (the per-row maximum over a few hundred columns is a noisy
    # statistic); a dropped split plane would show as 2^-16 = 1.5e-5 of the scale
    assert bool((e6 <= 2.0 * e32 + 1e-6 * rowscale).all()), (e6 / rowscale, e32 / rowscale)


@pytest.mark.parametrize("B,T,nslab,lens,with_pre", [(1, 384, 8, [384], True), (2, 77, 4, [77, 41], True), (3, 9, 1, [9, 1, 4], False),
                                                      (1, 1000, 2, [1000], True)])
def test_flow_boundary_vs_fp64(B, T, nslab, lens, with_pre):
    lib = _lib()
    Cc, C1 = 192, 96
    g = torch.Generator().manual_seed(T * 7 + B + nslab)
    slabs = torch.randn(nslab, B, Cc, T, generator=g)
    gamma, beta = 1 + 0.1 * torch.randn(Cc, generator=g), 0.1 * torch.randn(Cc, generator=g)
    z = torch.randn(B, 2 * C1, T, generator=g)          # x1 = rows [C1, 2 C1) of z
    pw, pb = torch.randn(C1, Cc, generator=g) / math.sqrt(Cc), 0.1 * torch.randn(C1, generator=g)
    qw, qb = torch.randn(Cc, C1, generator=g) / math.sqrt(C1), 0.1 * torch.randn(Cc, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()
    a = slabs.double().sum(0)
    mu = a.mean(1, keepdim=True)
    var = ((a - mu) ** 2).mean(1, keepdim=True)
    h = ((a - mu) / torch.sqrt(var + 1e-5) * gamma.double()[None, :, None] + beta.double()[None, :, None]) * mask.double()[:, None, :]
    x1 = (z[:, C1:].double() - (torch.einsum("oc,bct->bot", pw.double(), h) + pb.double()[None, :, None])) * mask.double()[:, None, :]
    pre = (torch.einsum("oc,bct->bot", qw.double(), x1) + qb.double()[None, :, None]) * mask.double()[:, None, :]
    sd, zd = slabs.cuda().contiguous(), z.cuda().contiguous()
    pre_out = torch.full((B, Cc, T), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_flow_boundary_pack_floats(Cc), device="cuda")
    x1d = zd.view(-1)[C1 * T:]                          # rows [C1, 2 C1) of item 0; the kernel strides by z_bstride per item
    gd, bd, md = gamma.cuda(), beta.cuda(), mask.cuda()  # named: a temporary's block would be handed to the next temporary
    rc = lib.bv2_test_flow_boundary(None, P(sd), nslab, B * Cc * T, P(gd), P(bd), P(md), P(x1d), 2 * C1 * T,
                                    P(pw.contiguous()), P(pb), P(qw.contiguous()) if with_pre else None, P(qb) if with_pre else None,
                                    P(pre_out) if with_pre else None, P(wp), B, Cc, T)
    assert rc == 0, rc
    torch.cuda.synchronize()
    got_x1 = zd[:, C1:].cpu().double()
    assert torch.equal(zd[:, :C1].cpu(), z[:, :C1])     # x0 is not touched
    s1 = x1.abs().max().item()
    assert (got_x1 - x1).abs().max().item() <= 2e-5 * max(s1, 1.0)
    if with_pre:
        s2 = pre.abs().max().item()
        assert (pre_out.cpu().double() - pre).abs().max().item() <= 2e-5 * max(s2, 1.0)
    # masked columns are exactly zero
    m0 = (mask == 0)[:, None, :]
    assert torch.equal(zd[:, C1:].cpu() * m0, torch.zeros(B, C1, T))
