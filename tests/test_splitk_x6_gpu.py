"""GPU: kernels/splitk_x6.hip — the split-K conv tile of the small-N regime (batch 1) with its products on the bf16 matrix core from exact
three-way bf16 splits (conv_x6.hip's arithmetic), for the Encoder FFN convolutions (reference attentions.py:438-446).  The claim is "fp32
accuracy": against fp64 the kernel is as close as the fp32 split-K kernel (conv_mfma.hip, tile 6) on the same inputs — on the real FFN
shapes (192 <-> 768, k = 3 / 5), with masks, ReLU, residual, K split into partial slabs, ragged tails.  (The product does not pack the
planes for these convs: measured, the kernel buys 0.4 % of a batch-1 step for +195 MB of blob — profiles/r05_ab_splitk_x6_not_kept.txt.)"""
import math

import pytest
import torch
import torch.nn.functional as F

from tests.test_kernels_gpu import _lib, P, rel_err

pytestmark = pytest.mark.gpu

TILE_SPLITK, TILE_SPLITK_X6 = 6, 13


def _run(lib, x, w, bias, tile, k, dil, relu=0, res=None, res_mode=0, in_mask=None, out_mask=None, mask_pre=0, mask_post=0, ksplit=1,
         lrelu=0.0):
    B, cin, L = x.shape
    cout = w.shape[0]
    slab = B * cout * L
    out = torch.full((ksplit, B, cout, L), float("nan"), device="cuda")
    wp = torch.empty(lib.bv2_test_conv_pack_floats(cin, cout, k), device="cuda")
    rc = lib.bv2_test_conv1d(None, P(x), P(w), P(bias), P(out), P(wp), B, cin, cout, k, dil, -1, L, tile, lrelu, relu, P(res), res_mode,
                             P(in_mask), P(out_mask), mask_pre, mask_post, None, 1, None, None, 1.0, ksplit, slab)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return out.sum(0)                                   # the consumer sums the partial slabs


CASES = [
    # B, cin, cout, k, dil, L, ksplit          (12 / 8 / 6 / 4 waves by what divides the K slice)
    (1, 192, 768, 5, 1, 384, 1),                # flow FFN conv_1: 12 groups -> 12 waves
    (1, 768, 192, 5, 1, 384, 4),                # flow FFN conv_2: 48 groups / 4 slabs -> 12 waves
    (1, 768, 192, 5, 1, 384, 8),                #                  / 8 slabs -> 6 waves
    (1, 768, 192, 3, 1, 128, 2),                # text encoder conv_2, 2 slabs: 24 groups -> 12 waves x 2 groups, 4 octets per wave
    (1, 192, 768, 3, 1, 128, 1),                # text encoder conv_1
    (2, 256, 256, 3, 1, 77, 1),                 # DurationPredictor-like: 16 groups -> 8 waves
    (1, 64, 96, 7, 3, 45, 1),                   # 4 groups -> 4 waves; dilation; 32 + 18 staged columns
    (3, 128, 40, 1, 1, 9, 2),                   # k = 1, cout not a multiple of 32, tiny L, 2 slabs
    (1, 192, 192, 5, 2, 200, 1),
]


@pytest.mark.parametrize("B,cin,cout,k,dil,L,ksplit", CASES)
def test_splitk_x6_is_as_accurate_as_the_fp32_splitk_kernel(B, cin, cout, k, dil, L, ksplit):
    lib = _lib()
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + k + L + ksplit)
    x = torch.randn(B, cin, L, generator=g) * torch.exp(2.0 * torch.randn(B, cin, 1, generator=g))     # a wide dynamic range in both operands
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k) * torch.exp(torch.randn(cout, 1, 1, generator=g))
    bias = torch.randn(cout, generator=g)
    ref = F.conv1d(x.double(), w.double(), bias.double(), padding=(k - 1) // 2 * dil, dilation=dil)
    xd = x.cuda()
    y6 = _run(lib, xd, w, bias, TILE_SPLITK_X6, k, dil, ksplit=ksplit)
    y32 = _run(lib, xd, w, bias, TILE_SPLITK, k, dil, ksplit=ksplit)
    e6, e32 = rel_err(y6, ref), rel_err(y32, ref)
    print(f"\n[{cin}->{cout} k={k} d={dil} L={L} slabs={ksplit}] rel err x6 {e6:.2e}, fp32 MFMA {e32:.2e}")
    assert e6 < 2e-5
    assert e6 <= 2.0 * e32 + 2e-7, (e6, e32)
    scale = ref.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)      # per output row: nothing hides behind the largest row
    r6 = ((y6.double().cpu() - ref).abs() / scale).max().item()
    r32 = ((y32.double().cpu() - ref).abs() / scale).max().item()
    assert r6 <= 2.0 * r32 + 4e-7, (r6, r32)


@pytest.mark.parametrize("res_mode", [0, 1, 2])
def test_splitk_x6_ffn_epilogues(res_mode):
    """The two FFN forms of attentions.py:438-446 at batch 2 with ragged lengths: conv_1 = relu(conv(x * mask)), conv_2 = conv(h * mask) * mask
    (+ residual), plus the reverse-subtract residual of the coupling layers."""
    lib = _lib()
    g = torch.Generator().manual_seed(5 + res_mode)
    B, cin, cout, k, L = 2, 192, 192, 5, 150
    x = torch.randn(B, cin, L, generator=g)
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias = torch.randn(cout, generator=g)
    res = torch.randn(B, cout, L, generator=g)
    mask = (torch.arange(L)[None, :] < torch.tensor([L, L - 61])[:, None]).float()
    md, xd, rd = mask.cuda(), x.cuda(), res.cuda()
    xin = x.double() * mask[:, None].double()
    y = F.conv1d(xin, w.double(), bias.double(), padding=(k - 1) // 2)
    # conv_1 form
    out = _run(lib, xd, w, bias, TILE_SPLITK_X6, k, 1, relu=1, in_mask=md)
    assert rel_err(out, torch.relu(y)) < 2e-5
    # conv_2 form
    ym = y * mask[:, None].double()
    ref = ym if res_mode == 0 else ((ym + res.double()) if res_mode == 1 else (res.double() - ym))
    out = _run(lib, xd, w, bias, TILE_SPLITK_X6, k, 1, res=rd if res_mode else None, res_mode=res_mode, in_mask=md, out_mask=md, mask_pre=1)
    assert rel_err(out, ref) < 2e-5


def test_unsupported_launches_fall_back_or_are_declined():
    lib = _lib()
    x = torch.randn(1, 24, 40).cuda()                   # 24 channels: not whole 16-channel groups
    w, b = torch.randn(32, 24, 3), torch.randn(32)
    out = torch.zeros(1, 32, 40, device="cuda")
    wp = torch.empty(lib.bv2_test_conv_pack_floats(24, 32, 3), device="cuda")
    rc = lib.bv2_test_conv1d(None, P(x), P(w), P(b), P(out), P(wp), 1, 24, 32, 3, 1, -1, 40, TILE_SPLITK_X6, 0.0, 0, None, 0, None, None, 0, 0,
                             None, 1, None, None, 1.0, 1, 0)
    assert rc != 0
