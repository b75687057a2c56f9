"""GPU: seeded random shapes against the oracle — lengths that are not multiples of any tile size, single-symbol and
single-frame utterances, ragged batches, both flow types, every precision switch, exact-length batching, max_len.
Each case: durations / alignment path exact, every [.,.,T_y] tensor and the waveform within the fp32 (or the mode's)
tolerance on the valid region."""
import random

import pytest
import torch

from bert_vits2_amd import hparams as H, synth
from oracle import bv2_oracle as O
from tests.helpers import cached_state_dict, rms, valid_wave_mask

pytestmark = pytest.mark.gpu


def _cases():
    rnd = random.Random(20260923)
    out = [dict(lengths=[1], tf=True), dict(lengths=[2, 1, 1], tf=True), dict(lengths=[33], tf=False), dict(lengths=[31, 32, 1, 17], tf=True)]
    for i in range(6):
        B = rnd.randint(1, 4)
        out.append(dict(lengths=[rnd.randint(1, 70) for _ in range(B)], tf=rnd.random() < 0.7))
    for i, c in enumerate(out):
        c["seed"] = i % 3
        c["sdp_ratio"] = [0.0, 0.5, 1.0][i % 3]
        c["length_scale"] = [1.0, 0.7, 1.3][(i // 3) % 3]
    return out


CASES = _cases()
_MODELS = {}


def _model(hp, seed):
    from bert_vits2_amd import models
    key = (hp.use_transformer_flow, seed)
    if key not in _MODELS:
        m = models.from_hparams(hp)
        m.load_state_dict(cached_state_dict(hp, seed), strict=False)
        _MODELS[key] = m.to("cuda").eval()
    return _MODELS[key]


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_random_shapes_vs_oracle(idx):
    c = CASES[idx]
    hp = H.default_v23(use_transformer_flow=c["tf"])
    sd = cached_state_dict(hp, c["seed"])
    B, T = len(c["lengths"]), max(c["lengths"])
    g = random.Random(idx)
    batch = synth.synthetic_batch(c["lengths"], [g.randint(0, 2) for _ in range(B)], [g.randint(0, 849) for _ in range(B)],
                                  first_index=10 * idx)
    nw, nz = synth.synthetic_noise(B, T, 12 * T + 64, hp.inter_channels, seed=100 + idx)
    kw = dict(noise_scale=0.6, noise_scale_w=0.9, sdp_ratio=c["sdp_ratio"], length_scale=c["length_scale"])
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    m = _model(hp, c["seed"])
    m.set_generator_dtype(torch.float32)
    m.set_flow_dtype(torch.float32)
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)
    torch.cuda.synchronize()
    # the durations the kernels computed themselves: exact up to rare ceil() flips at fp32 round-off distance
    flips = (m.last_encode["logw"].cpu()[:, None] - ref["logw"]).abs().max().item()
    assert flips < 1e-4, flips
    assert torch.equal(attn.cpu(), ref["attn"]) and torch.equal(y_mask.cpu(), ref["y_mask"])
    Ty = y_mask.shape[2]
    ym = ref["y_mask"]
    for got, key in ((z_p, "z_p"), (m_p, "m_p"), (logs_p, "logs_p"), (z, "z")):
        assert rms((got.cpu() - ref[key]) * ym) <= 2e-5 * max(rms(ref[key] * ym), 1e-3), key
    vm = valid_wave_mask(ref["y_lengths"], hp.total_upsample, o.shape[2]).expand_as(ref["o"])
    assert rms((o.cpu() - ref["o"])[vm]) <= 5e-5
    # max_len (models.py:1073): a shorter decode equals the prefix of a decoder run on the truncated latent
    if Ty >= 4:
        Lh = Ty // 2
        o2 = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], max_len=Lh, **kw)[0]
        ref2 = O.generator(sd, hp, (ref["z"] * ym)[:, :, :Lh], ref["g"])
        assert o2.shape == ref2.shape and rms(o2.cpu() - ref2) <= 5e-5
    # exact lengths: every utterance of the padded batch equals its own single-utterance oracle run
    if B > 1:
        oe = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], exact_lengths=True, **kw)[0].cpu()
        for b in range(B):
            n = c["lengths"][b]
            one = {k: batch[k][b:b + 1, ..., :n] if batch[k].dim() > 1 else batch[k][b:b + 1] for k in batch}
            one["x_lengths"] = batch["x_lengths"][b:b + 1]
            wc = ref["w_ceil"][b:b + 1, :, :n]
            r1 = O.infer(sd, hp, one["x"], one["x_lengths"], one["sid"], one["tone"], one["language"], one["bert"], one["ja_bert"],
                         one["en_bert"], noise_w=nw[b:b + 1, :, :n], noise_z=nz[b:b + 1], w_ceil_override=wc, **kw)
            S1 = r1["o"].shape[2]
            assert rms(oe[b, 0, :S1] - r1["o"][0, 0]) <= 5e-5, (b, rms(oe[b, 0, :S1] - r1["o"][0, 0]))
    # reduced-precision switches: finite, same path, waveform inside the mode's tolerance
    m.set_flow_dtype(torch.float16)                  # both flow variants (transformer / WN) have an fp16 form
    o16, attn16, _, (z16, *_rest) = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)
    assert torch.equal(attn16.cpu(), ref["attn"]) and torch.isfinite(o16).all()
    assert rms((z16.cpu() - ref["z"]) * ym) <= 5e-3 * max(rms(ref["z"] * ym), 1e-3)
    assert rms((o16.cpu() - ref["o"])[vm]) <= 1e-3
    m.set_generator_dtype(torch.bfloat16)
    ob = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)[0]
    assert torch.isfinite(ob).all()
    assert rms((ob.cpu() - ref["o"])[vm]) <= 5e-2 * max(rms(ref["o"][vm]), 1e-2)
    m.set_generator_dtype(torch.float32)
    m.set_flow_dtype(torch.float32)
