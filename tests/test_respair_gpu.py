"""GPU: the pair-fused bf16 ResBlock kernel of the wide Generator stages (kernels/respair_cl_bf16.hip: dilated conv -> LDS -> conv +
residual in one launch, C = 256 / 128 / 64) against the layer-wise path it replaces (kernels/gen_bf16.hip, two launches per pair,
reference modules.py:296-309).  Both run the same GEMM unit order and round at the same points, so they must agree BIT FOR BIT —
every ResBlock output of every stage (debug taps) and the waveform; the layer-wise path itself is held to the oracle and the
reference goldens by tests/test_bf16_gpu.py."""
import pytest
import torch

from oracle import cases
from tests.helpers import cached_state_dict, valid_wave_mask

pytestmark = pytest.mark.gpu


def _gpu_model(hp, seed):
    from bert_vits2_amd import models
    m = models.from_hparams(hp)
    m.load_state_dict(cached_state_dict(hp, seed), strict=False)
    m = m.to("cuda").eval()
    m.set_generator_dtype(torch.bfloat16)
    return m


def _run_with_taps(m, hp, z, yl, g, fused, form=1):
    m.set_option("fused_respair", fused)
    m.set_option("respair_form", form)
    B, _, Ty = z.shape
    taps, up = {}, 1
    for i, u in enumerate(hp.upsample_rates):
        up *= u
        ch = hp.upsample_initial_channel // 2 ** (i + 1)
        for j in range(3):
            taps[f"dec.rb.{i}.{j}"] = torch.full((B, ch, Ty * up), float("nan"), device="cuda")
    for k, t in taps.items():
        m.set_tap(k, t)
    try:
        o = m.stage_generator(z, yl, g)
        torch.cuda.synchronize()
    finally:
        m.set_tap(None)
    return o, taps


@pytest.mark.parametrize("B,Ty,lens", [(2, 300, [300, 177]), (1, 37, [37]), (3, 129, [129, 5, 64]), (1, 700, [700])])
def test_pair_fused_equals_layer_wise_bit_for_bit(B, Ty, lens):
    hp, seed, *_ = cases.build_case("zh_b1_t24")
    m = _gpu_model(hp, seed)
    gen = torch.Generator().manual_seed(B * 1000 + Ty)
    z = torch.randn(B, hp.inter_channels, Ty, generator=gen).cuda()
    g = torch.randn(B, hp.gin_channels, 1, generator=gen).cuda()
    yl = torch.tensor(lens, dtype=torch.int64).cuda()
    o0, t0 = _run_with_taps(m, hp, z, yl, g, 0)
    assert torch.isfinite(o0).all()
    # both forms of the pair kernel: 64-channel x 128-row wave tiles on the swizzled tile (1, the default), 32-channel waves (0)
    for form in (1, 0):
        o1, t1 = _run_with_taps(m, hp, z, yl, g, 1, form)
        assert torch.isfinite(o1).all()
        for k in t0:
            assert torch.isfinite(t1[k]).all(), (form, k)
            assert torch.equal(t1[k], t0[k]), (form, k, (t1[k] - t0[k]).abs().max().item(), (t1[k] != t0[k]).float().mean().item())
        assert torch.equal(o1, o0), form


def test_pair_fused_exact_lengths_and_graph_replay():
    """The exact-length form (every utterance of a padded batch ends at ITS length: rows past it are the convs' zero padding) and the
    hipGraph replay of phase B go through the pair kernel too: valid samples equal the layer-wise path's."""
    name = "mix_b2_ragged"
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    m = _gpu_model(hp, seed)
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    outs = {}
    for fused in (1, 0):
        m.set_option("fused_respair", fused)
        o, _, y_mask, _ = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), exact_lengths=True, **kw)
        outs[fused] = (o.cpu(), y_mask.sum((1, 2)).long().cpu())
    vm = valid_wave_mask(outs[1][1], hp.total_upsample, outs[1][0].shape[2]).expand_as(outs[1][0])
    assert torch.equal(outs[1][0][vm], outs[0][0][vm])
    m.set_option("fused_respair", 1)
    m.enable_graphs(True, ty_bucket=1)
    try:
        for _ in range(2):
            og = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), exact_lengths=True, **kw)[0].cpu()
        assert torch.equal(og[vm], outs[1][0][vm])
    finally:
        m.enable_graphs(False)


def test_pair_kernel_on_the_c32_stage_equals_layer_wise():
    """The C = 32 stage pair by pair (one wave owns all 32 channels, 512-row tiles; the default) against the same stage one conv per
    launch."""
    hp, seed, *_ = cases.build_case("zh_b1_t24")
    m = _gpu_model(hp, seed)
    gen = torch.Generator().manual_seed(321)
    B, Ty = 2, 150
    z = torch.randn(B, hp.inter_channels, Ty, generator=gen).cuda()
    g = torch.randn(B, hp.gin_channels, 1, generator=gen).cuda()
    yl = torch.tensor([150, 83], dtype=torch.int64).cuda()
    m.set_option("fused_resblock", 0)
    try:
        o0, t0 = _run_with_taps(m, hp, z, yl, g, 0)
        m.set_option("fused_resblock", 1)
        o1, t1 = _run_with_taps(m, hp, z, yl, g, 1)
    finally:
        m.set_option("fused_resblock", 1)
    # the layer-wise C = 32 kernel walks its K dimension group-major (generic loop), the pair kernel tap-major: same rounding points, a
    # different fp32 summation order — outputs agree except for rare one-ulp bf16 flips (2^-8 relative) that propagate through the pairs
    for k in [f"dec.rb.3.{j}" for j in range(3)]:
        a, b = t1[k].double(), t0[k].double()
        assert torch.isfinite(t1[k]).all()
        scale = b.abs().max().item()
        diff = (a - b).abs()
        frac = (diff > 0).double().mean().item()
        print(f"\n[{k}] pair vs layer-wise: max |diff| {diff.max().item():.3e} (scale {scale:.3e}), differing elements {frac:.2e}")
        assert diff.max().item() <= 2.0 ** -5 * scale and frac < 0.05, (k, diff.max().item(), scale, frac)
        assert (diff.pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item() < 2e-3
    assert (o1 - o0).pow(2).mean().sqrt().item() <= 1e-2 * o0.pow(2).mean().sqrt().item()
