"""GPU: the attention kernel's fused output projection (conv_o inside the attention workgroup, head h -> partial slab h summed by the
LayerNorm that follows; kernels/attention.hip) against the two-launch form (attention, then the split-K conv_o) and the oracle.
Reference: attentions.py:262-271 (MultiHeadAttention.forward) + :118-121 (Encoder: x = norm_layers_1(x + y))."""
import pytest
import torch

from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["mix_b2_ragged", "mid_b2_t72", "t1_b1", "t4_b2", "short_b3"])
def test_fused_conv_o_matches_separate_launch_and_oracle(name):
    from bert_vits2_amd import models
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, want_taps=True, **kw)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()

    def run():
        enc = m.encode_durations(batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                                 batch["ja_bert"], batch["en_bert"], nw, noise_scale_w=kw["noise_scale_w"],
                                 sdp_ratio=kw["sdp_ratio"], length_scale=kw["length_scale"])
        # pinned durations (the oracle's): both forms decode the same frames
        o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"],
                                                         batch["bert"], batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz,
                                                         w_ceil=ref["w_ceil"], **kw)
        torch.cuda.synchronize()
        return {k: enc[k].clone() for k in ("x", "m_p", "logs_p")}, dict(o=o.clone(), z=z.clone())

    e1, d1 = run()
    m.set_option("fused_attn_o", 0)
    e0, d0 = run()
    m.set_option("fused_attn_o", 1)
    # text encoder (6 layers) and flow (4 x 4 layers): the two forms differ by fp32 summation order only
    for k in ("x", "m_p", "logs_p"):
        s = e0[k].abs().max().item()
        assert (e1[k] - e0[k]).abs().max().item() <= 2e-5 * max(s, 1.0), k
        assert ((e1[k].cpu() - ref["enc_" + k.replace("_p", "")]).abs().max() / ref["enc_" + k.replace("_p", "")].abs().max()).item() < 3e-4, k
    for k in ("z", "o"):
        s = d0[k].abs().max().item()
        assert (d1[k] - d0[k]).abs().max().item() <= 1e-4 * max(s, 1.0), (k, (d1[k] - d0[k]).abs().max().item(), s)
    assert ((d1["z"].cpu() - ref["z"]) * ref["y_mask"]).abs().max().item() < 1e-3
