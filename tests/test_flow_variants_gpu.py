"""GPU: the flow reverse (reference models.py:138-145) for hyper-parameters OTHER than the released config's, one deviation at a time, against
the oracle's flow_reverse.  Round 5: this bisection found the odd-coupling-count bug (the folded channel Flips cancel only for an even number of
couplings: n_flow_layer = 3 came out channel-reversed, rel. error 1.4; every other deviation was at 3e-7) — see bv2_model.cpp `flow_flip_first`.
The narrow model of tests/golden/narrow_b2_t18.npz holds all deviations at once against the REAL reference."""
import pytest
import torch

from oracle import bv2_oracle as O
from tests.helpers import rms

pytestmark = pytest.mark.gpu

VARIANTS = {
    "n_flow_layer=3": dict(n_flow_layer=3),
    "n_flow_layer=2": dict(n_flow_layer=2),
    "n_layers_trans_flow=3": dict(n_layers_trans_flow=3),
    "hidden=128,filter=512": dict(hidden_channels=128, filter_channels=512),
    "inter=128": dict(inter_channels=128),
    "gin=256": dict(gin_channels=256),
}


@pytest.mark.parametrize("name", list(VARIANTS))
def test_flow_reverse_with_one_hyperparameter_off_the_released_config(name):
    from bert_vits2_amd import hparams as H, models, synth
    hp = H.default_v23(**VARIANTS[name])
    sd = synth.synthetic_state_dict(hp, seed=5)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    B, Ty = 2, 45
    gen = torch.Generator().manual_seed(3)
    yl = torch.tensor([45, 29], dtype=torch.int64)
    ym = (torch.arange(Ty)[None, :] < yl[:, None])[:, None, :].float()
    z_p = torch.randn(B, hp.inter_channels, Ty, generator=gen) * ym
    g = torch.randn(B, hp.gin_channels, 1, generator=gen)
    ref = O.flow_reverse(sd, hp, z_p, ym, g)
    for fb in (1, 0):                                  # with and without the fused boundary launch (declined where the widths differ)
        m.set_option("fused_boundary", fb)
        z = m.stage_flow(z_p.cuda(), yl.cuda(), g.cuda()).cpu()
        e = rms((z - ref) * ym) / max(rms(ref * ym), 1e-3)
        print(f"\n[{name}] fused_boundary={fb}: rel RMS error of z vs oracle {e:.3e}")
        assert e <= 2e-5, (name, fb, e)
