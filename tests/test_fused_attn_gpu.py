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


@pytest.mark.parametrize("lengths,pin", [([128], 2.5), ([128, 77], None), ([40], None)])
def test_key_split_attention_matches_unsplit_and_oracle(lengths, pin):
    """Key split (kernels/attention.hip, AttnArgs::ksplit): the key tiles of a (head, query tile) are dealt to 2 / 4 workgroups whose
    partial slabs LayerNorm-1 merges with the flash-decoding weights.  Same result as the unsplit kernel up to fp32 summation order,
    for the shape the picker splits by itself (T_y = 384 at batch 1: 12 key tiles -> 4 ranges), a ragged batch of two (masked keys
    inside a range, an utterance that ends mid-range) and a short utterance where the picker must stay at 1."""
    from bert_vits2_amd import hparams as H, models, synth
    hp = H.default_v23()
    sd = cached_state_dict(hp, 0, **({} if pin is None else dict(pin_durations=pin)))
    batch = synth.synthetic_batch(lengths, [0, 2][: len(lengths)], [0, 7][: len(lengths)])
    B, T = batch["x"].shape
    nw, nz = synth.synthetic_noise(B, T, 3 * T + 64 if pin else 24 * T, hp.inter_channels)
    kw = dict(noise_scale=0.6, noise_scale_w=0.9, sdp_ratio=0.0 if pin else 0.5, length_scale=1.0)
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]

    def run(ks):
        m.set_option("attn_ksplit", ks)
        o, attn, y_mask, (z, *_r) = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)
        torch.cuda.synchronize()
        return o.clone(), z.clone()

    o1, z1 = run(0)
    valid = ref["y_mask"].cuda()
    for ks in (-1, 2, 4):
        o, z = run(ks)
        assert o.shape == o1.shape
        sz, so = z1.abs().max().item(), o1.abs().max().item()
        assert ((z - z1) * valid).abs().max().item() <= 1e-4 * max(sz, 1.0), (ks, ((z - z1) * valid).abs().max().item())
        assert (o - o1).abs().max().item() <= 2e-4 * max(so, 1.0), (ks, (o - o1).abs().max().item())
        assert ((z.cpu() - ref["z"]) * ref["y_mask"]).abs().max().item() < 1e-3, ks
        n = int(ref["y_lengths"].min()) * hp.total_upsample
        err = (o.cpu()[:, :, :n] - ref["o"][:, :, :n]).pow(2).mean().sqrt().item()
        assert err < 5e-5, (ks, err)
    m.set_option("attn_ksplit", -1)
