"""GPU: the fused DDSConv-layer kernel (kernels/dds_fused.hip) against a torch fp64 restatement of reference modules.py:121-129 (+ the
ConvFlow.pre input transform, modules.py:488-489, and the projection / inverse spline that follows the DDSConv, modules.py:491-516),
and the fused stochastic duration predictor end to end against the layer-wise kernels and the oracle."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict

pytestmark = pytest.mark.gpu


def _lib():
    from bert_vits2_amd import lib as L
    lib = L.load()
    lib.bv2_test_dds_pack_floats.restype = C.c_int64
    lib.bv2_test_dds_pack_floats.argtypes = [C.c_int]
    lib.bv2_test_dds_layer.restype = C.c_int
    lib.bv2_test_dds_layer.argtypes = ([C.c_void_p] * 5 + [C.c_int] + [C.c_void_p] * 11 + [C.c_int, C.c_int] + [C.c_void_p] * 2 +
                                       [C.c_int] + [C.c_void_p] * 2 + [C.c_int, C.c_void_p] + [C.c_int] * 3)
    return lib


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def ln_c(x, g, b):
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), g, b, 1e-5).transpose(1, 2)


def ref_layer(xin, mask, dww, dwb, g1, b1, w, bias, g2, b2, dil, last):
    C_ = xin.shape[1]
    y = F.conv1d(xin * mask, dww[:, None, :], dwb, padding=dil, dilation=dil, groups=C_)
    y = F.gelu(ln_c(y, g1, b1))
    y = F.conv1d(y, w[:, :, None], bias)
    y = F.gelu(ln_c(y, g2, b2))
    out = xin + y
    return out * mask if last else out


def _params(Cc, g):
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    return dict(dww=r(Cc, 3, sc=0.6), dwb=r(Cc, sc=0.1), g1=1 + r(Cc, sc=0.1), b1=r(Cc, sc=0.1), g2=1 + r(Cc, sc=0.1), b2=r(Cc, sc=0.1),
                w=r(Cc, Cc, sc=1 / math.sqrt(Cc)), bias=r(Cc, sc=0.1))


@pytest.mark.parametrize("Cc,B,T,dil,lens,last", [(192, 1, 128, 1, [128], 0), (192, 2, 77, 3, [77, 40], 1), (192, 3, 50, 9, [50, 9, 1], 1),
                                                   (128, 2, 33, 3, [33, 20], 0), (256, 1, 40, 1, [31], 1), (192, 1, 5, 9, [5], 1)])
def test_dds_layer_plain(Cc, B, T, dil, lens, last):
    lib = _lib()
    g = torch.Generator().manual_seed(Cc + T + dil)
    p = _params(Cc, g)
    x = torch.randn(B, Cc, T, generator=g)
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()[:, None]
    d = {k: v.double() for k, v in p.items()}
    ref = ref_layer(x.double(), mask.double(), d["dww"], d["dwb"], d["g1"], d["b1"], d["w"], d["bias"], d["g2"], d["b2"], dil, last)
    xd, md = x.cuda(), mask[:, 0].contiguous().cuda()
    out = torch.full((B, Cc, T), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_dds_pack_floats(Cc), device="cuda")
    rc = lib.bv2_test_dds_layer(None, P(xd), None, None, None, 0, None, P(md), P(p["dww"]), P(p["dwb"]), P(p["g1"]), P(p["b1"]),
                                P(p["g2"]), P(p["b2"]), P(p["w"]), P(p["bias"]), P(out), dil, last, None, None, 0, None, None, 0,
                                P(wp), B, Cc, T)
    assert rc == 0
    torch.cuda.synchronize()
    err = ((out.double().cpu() - ref).abs().max() / ref.abs().max()).item()
    assert err < 2e-5, err


def test_dds_layer_convflow_pre_and_spline_epilogue():
    """Layer 0 fed by ConvFlow.pre (h = w*z[src] + b, then + g), and a closing layer whose 29-row projection parametrises the
    inverse spline applied to z[dst] — one ConvFlow of the stochastic duration predictor, layer by layer."""
    lib = _lib()
    Cc, B, T = 192, 2, 45
    g = torch.Generator().manual_seed(9)
    lens = [45, 30]
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()[:, None]
    z = torch.randn(B, 2, T, generator=g) * 1.5
    z[0, 0, 3] = 6.5                                           # outside the +-5 tail: identity branch
    cond = torch.randn(B, Cc, T, generator=g) * 0.5
    pre_w, pre_b = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g) * 0.1
    pA = _params(Cc, g)
    src, dst = 1, 0
    # layer 0 with the pre transform
    xin = (pre_w[None, :, None] * z[:, src:src + 1] + pre_b[None, :, None] + cond).double()
    d = {k: v.double() for k, v in pA.items()}
    ref0 = ref_layer(xin, mask.double(), d["dww"], d["dwb"], d["g1"], d["b1"], d["w"], d["bias"], d["g2"], d["b2"], 1, 0)
    md = mask[:, 0].contiguous().cuda()
    out0 = torch.full((B, Cc, T), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_dds_pack_floats(Cc), device="cuda")
    zd, cd = z.cuda(), cond.cuda()
    rc = lib.bv2_test_dds_layer(None, None, P(pre_w), P(pre_b), P(zd), src, P(cd), P(md), P(pA["dww"]), P(pA["dwb"]), P(pA["g1"]),
                                P(pA["b1"]), P(pA["g2"]), P(pA["b2"]), P(pA["w"]), P(pA["bias"]), P(out0), 1, 0, None, None, 0, None,
                                None, 0, P(wp), B, Cc, T)
    assert rc == 0
    torch.cuda.synchronize()
    assert ((out0.double().cpu() - ref0).abs().max() / ref0.abs().max()).item() < 2e-5
    # closing layer + projection + spline (out = null: only z is updated)
    pB = _params(Cc, g)
    proj_w = torch.randn(29, Cc, generator=g) * (3.0 / math.sqrt(Cc))
    proj_b = torch.randn(29, generator=g) * 0.1
    x1 = torch.randn(B, Cc, T, generator=g)
    d = {k: v.double() for k, v in pB.items()}
    h = ref_layer(x1.double(), mask.double(), d["dww"], d["dwb"], d["g1"], d["b1"], d["w"], d["bias"], d["g2"], d["b2"], 9, 1)
    prm = (F.conv1d(h, proj_w.double()[:, :, None], proj_b.double()) * mask.double()).transpose(1, 2)     # [B,T,29]
    uw, uh, ud = prm[..., :10] / math.sqrt(Cc), prm[..., 10:20] / math.sqrt(Cc), prm[..., 20:]
    zref = z.double().clone()
    zref[:, dst] = O.rq_spline_inverse(zref[:, dst], uw, uh, ud, 5.0) * mask.double()[:, 0]
    zref[:, src] = zref[:, src] * mask.double()[:, 0]
    zio = z.clone().cuda()
    rc = lib.bv2_test_dds_layer(None, P(x1.cuda()), None, None, None, src, None, P(md), P(pB["dww"]), P(pB["dwb"]), P(pB["g1"]),
                                P(pB["b1"]), P(pB["g2"]), P(pB["b2"]), P(pB["w"]), P(pB["bias"]), None, 9, 1, P(proj_w), P(proj_b), 29,
                                None, P(zio), dst, P(wp), B, Cc, T)
    assert rc == 0
    torch.cuda.synchronize()
    assert (zio.double().cpu() - zref).abs().max().item() < 2e-4        # spline amplifies fp32 round-off of its parameters
    assert zio[0, 0, 3].item() == pytest.approx(6.5)                    # tail: identity


def test_dds_layer_post_projection_rows():
    """The closing layer of the SDP trunk: post_out = (W_proj h + b) * mask over all C rows (sdp.proj, models.py:203-204)."""
    lib = _lib()
    Cc, B, T = 192, 2, 70
    g = torch.Generator().manual_seed(21)
    lens = [70, 51]
    mask = (torch.arange(T)[None, :] < torch.tensor(lens)[:, None]).float()[:, None]
    p = _params(Cc, g)
    pw, pb = torch.randn(Cc, Cc, generator=g) / math.sqrt(Cc), torch.randn(Cc, generator=g) * 0.1
    x = torch.randn(B, Cc, T, generator=g)
    d = {k: v.double() for k, v in p.items()}
    h = ref_layer(x.double(), mask.double(), d["dww"], d["dwb"], d["g1"], d["b1"], d["w"], d["bias"], d["g2"], d["b2"], 9, 1)
    ref = F.conv1d(h, pw.double()[:, :, None], pb.double()) * mask.double()
    md = mask[:, 0].contiguous().cuda()
    post = torch.full((B, Cc, T), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_dds_pack_floats(Cc), device="cuda")
    rc = lib.bv2_test_dds_layer(None, P(x.cuda()), None, None, None, 0, None, P(md), P(p["dww"]), P(p["dwb"]), P(p["g1"]), P(p["b1"]),
                                P(p["g2"]), P(p["b2"]), P(p["w"]), P(p["bias"]), None, 9, 1, P(pw), P(pb), Cc, P(post), None, 0, P(wp),
                                B, Cc, T)
    assert rc == 0
    torch.cuda.synchronize()
    assert ((post.double().cpu() - ref).abs().max() / ref.abs().max()).item() < 2e-5


@pytest.mark.parametrize("name", ["mix_b2_ragged", "mid_b2_t72", "t1_b1"])
def test_fused_predictor_matches_layerwise_kernels_and_oracle(name):
    from bert_vits2_amd import models
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    run = lambda: m.encode_durations(batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                                     batch["ja_bert"], batch["en_bert"], nw, noise_scale_w=kw["noise_scale_w"],
                                     sdp_ratio=kw["sdp_ratio"], length_scale=kw["length_scale"])
    fused = run()
    m.set_option("fused_dds", 0)
    layerwise = run()
    m.set_option("fused_dds", 1)
    torch.cuda.synchronize()
    for k in ("logw_sdp", "logw_dp", "logw"):
        assert (fused[k] - layerwise[k]).abs().max().item() < 2e-4, k
        assert (fused[k].cpu() - ref[k][:, 0]).abs().max().item() < 1e-3, k
    assert (fused["w_ceil"] != layerwise["w_ceil"]).float().mean().item() <= 0.01
