"""GPU: kernels/flow_boundary.hip — LayerNorm-2 of a coupling's last Encoder layer, the coupling's `post` and the next coupling's `pre`
(reference attentions.py:118-120, models.py:121-132) in ONE launch, against the three-launch form it replaces (LayerNorm kernel + two
split-K 1x1 convs on the matrix core) and against the oracle.  Same LayerNorm statistics bit for bit; the 1x1 convs run on the VALU in a
different summation order, so the flow output agrees to fp32 round-off, not bit for bit."""
import pytest
import torch

from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict, rms

pytestmark = pytest.mark.gpu


def _gpu_model(hp, seed):
    from bert_vits2_amd import models
    m = models.from_hparams(hp)
    m.load_state_dict(cached_state_dict(hp, seed), strict=False)
    return m.to("cuda").eval()


@pytest.mark.parametrize("B,Ty,lens", [(1, 384, [384]), (2, 77, [77, 41]), (3, 9, [9, 1, 4]), (1, 1000, [1000])])
def test_fused_boundary_equals_three_launches(B, Ty, lens):
    hp, seed, *_ = cases.build_case("zh_b1_t24")
    m = _gpu_model(hp, seed)
    gen = torch.Generator().manual_seed(Ty + B)
    z_p = torch.randn(B, hp.inter_channels, Ty, generator=gen).cuda()
    g = torch.randn(B, hp.gin_channels, 1, generator=gen).cuda()
    yl = torch.tensor(lens, dtype=torch.int64).cuda()
    m.set_option("fused_boundary", 0)
    z0 = m.stage_flow(z_p, yl, g)
    m.set_option("fused_boundary", 1)
    z1 = m.stage_flow(z_p, yl, g)
    torch.cuda.synchronize()
    assert torch.isfinite(z1).all()
    scale = z0.abs().max().item()
    err = (z1 - z0).abs().max().item()
    print(f"\n[B={B} Ty={Ty}] fused boundary vs three launches: max |diff| {err:.3e} at scale {scale:.3e}")
    assert err <= 2e-5 * max(scale, 1.0), (err, scale)
    # padded frames stay exactly zero (the masks are applied inside the fused launch)
    mask = (torch.arange(Ty, device="cuda")[None, :] < yl[:, None])[:, None, :]
    assert torch.equal(z1 * (~mask), torch.zeros_like(z1))


@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged"])
def test_flow_with_fused_boundary_vs_oracle(name):
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    m = _gpu_model(hp, seed)
    z = m.stage_flow(ref["z_p"], ref["y_lengths"], ref["g"])
    ym = ref["y_mask"]
    assert rms((z.cpu() - ref["z"]) * ym) <= 2e-5 * max(rms(ref["z"] * ym), 1e-3)


def test_inter_channels_not_equal_hidden_takes_the_three_launch_form():
    """the fused kernel is built for x1 rows = hidden/2 = 96.  A config with hidden_channels = 192 and inter_channels = 256
    (legal in the reference: TransformerCouplingLayer.pre is half -> hidden, models.py:95-132) must NOT take it: the executor's gate
    (half * 2 == H) and flow_boundary_supported (C1 * 2 == C) both decline, and the flow agrees with the oracle."""
    from bert_vits2_amd import hparams as H, synth
    hp = H.default_v23(inter_channels=256)
    sd = synth.synthetic_state_dict(hp, seed=5)
    from bert_vits2_amd import models
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    B, Ty = 2, 77
    gen = torch.Generator().manual_seed(3)
    yl = torch.tensor([77, 41], dtype=torch.int64)
    ym = (torch.arange(Ty)[None, :] < yl[:, None])[:, None, :].float()
    z_p = torch.randn(B, hp.inter_channels, Ty, generator=gen) * ym
    g = torch.randn(B, hp.gin_channels, 1, generator=gen)
    m.set_option("fused_boundary", 0)
    z0 = m.stage_flow(z_p.cuda(), yl.cuda(), g.cuda())
    m.set_option("fused_boundary", 1)
    z1 = m.stage_flow(z_p.cuda(), yl.cuda(), g.cuda())
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)                       # same launches either way: the fused kernel was declined
    ref = O.flow_reverse(sd, hp, z_p, ym, g)
    assert rms((z1.cpu() - ref) * ym) <= 2e-5 * max(rms(ref * ym), 1e-3)
