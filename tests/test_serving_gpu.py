"""GPU: batched serving glue — a padded, length-bucketed batch gives every utterance the audio it gets on its own, and
the device-side 16-bit conversion equals the host conversion the reference's callers run (gradio convert_to_16_bit_wav)."""
import numpy as np
import pytest
import torch

from bert_vits2_amd import hparams as H, serving, synth
from tests.helpers import cached_state_dict

pytestmark = pytest.mark.gpu


def _model(hp, seed=0):
    from bert_vits2_amd import models
    m = models.from_hparams(hp)
    m.load_state_dict(cached_state_dict(hp, seed), strict=False)
    return m.to("cuda").eval()


def _utts(lengths):
    out = []
    for i, T in enumerate(lengths):
        b = synth.synthetic_batch([T], languages=[i % 3], sids=[i * 7 % 50], first_index=i)
        out.append(serving.Utterance(b["x"][0], b["tone"][0], b["language"][0], b["bert"][0], b["ja_bert"][0], b["en_bert"][0],
                                     int(b["sid"][0])))
    return out


@pytest.mark.parametrize("mode", ["fp32", "bf16+f16"])
def test_batched_equals_one_by_one(mode):
    hp = H.default_v23()
    m = _model(hp)
    if mode != "fp32":
        m.set_generator_dtype(torch.bfloat16)
        m.set_flow_dtype(torch.float16)
    lengths = [17, 24, 9, 22, 40, 20]
    utts = _utts(lengths)
    g = torch.Generator().manual_seed(5)
    noise = [(torch.randn(2, T, generator=g), torch.randn(hp.inter_channels, 16 * T, generator=g)) for T in lengths]
    kw = dict(sdp_ratio=0.5, noise_scale=0.6, noise_scale_w=0.9, length_scale=1.0)
    single = [serving.synthesize(m, [u], noise=[n], **kw)[0] for u, n in zip(utts, noise)]
    batched = serving.synthesize(m, utts, noise=noise, max_batch=4, max_pad_ratio=1.5, **kw)
    assert len(batched) == len(utts)
    for a, b in zip(single, batched):
        assert a.shape == b.shape and a.dtype == np.float32 and a.size > 0      # same durations -> same length
        # fp32: only the summation order differs (other tilings at other shapes); reduced precision: the same rounding points,
        # so only rare 1-ulp flips that the following layers propagate
        tol = 1e-5 if mode == "fp32" else 2e-2
        assert np.sqrt(np.mean((a - b) ** 2)) <= tol * max(np.sqrt(np.mean(a ** 2)), 1e-3)
    # without exact lengths the padded batch follows the reference's unmasked decoder: the tail of a short utterance differs
    b0 = serving.collate([utts[2], utts[4]], "cuda")
    nw = torch.zeros(2, 2, 40); nz = torch.zeros(2, hp.inter_channels, 16 * 40)
    for r, i in enumerate((2, 4)):
        nw[r, :, :lengths[i]] = noise[i][0]; nz[r, :, :16 * lengths[i]] = noise[i][1]
    o, _, ym, _ = m.infer(b0["x"], b0["x_lengths"], b0["sid"], b0["tone"], b0["language"], b0["bert"], b0["ja_bert"], b0["en_bert"],
                          noise_w=nw.cuda(), noise_z=nz.cuda(), **kw)
    n = int(ym[0].sum()) * hp.total_upsample
    ref_tail = single[2][-2048:]
    assert np.abs(o[0, 0, :n].cpu().numpy()[-2048:] - ref_tail).max() > 10 * np.abs(batched[2][-2048:] - ref_tail).max()


def test_pcm16_matches_host_conversion():
    hp = H.default_v23()
    m = _model(hp)
    g = torch.Generator().manual_seed(11)
    B, S = 3, 5 * hp.total_upsample + 7
    wave = (torch.randn(B, 1, S, generator=g) * 0.3).cuda()
    y_len = torch.tensor([5, 2, 0])
    wave[2] = 0
    pcm = serving.pcm16(m, wave, y_len).cpu().numpy()
    for b in range(B):
        n = int(y_len[b]) * hp.total_upsample
        x = wave[b, 0, :n].cpu().numpy()
        if n:
            ref = (x / np.abs(x).max() * 32767).astype(np.int16)                 # gradio convert_to_16_bit_wav
            # integer output: bit-exact.  The device kernel does the host's fp32 op order, (x / peak) * 32767 with an IEEE
            # (correctly rounded) division and truncation toward zero
            assert np.array_equal(pcm[b, :n], ref), int(np.abs(pcm[b, :n].astype(np.int32) - ref.astype(np.int32)).max())
        assert not pcm[b, n:].any()
    out = serving.synthesize(m, _utts([12, 14]), as_pcm16=True)
    assert all(o.dtype == np.int16 and np.abs(o).max() >= 32766 for o in out)


@pytest.mark.parametrize("mode", ["fp32", "bf16+f16"])
def test_requests_in_flight_do_not_change_the_audio(mode):
    """Buckets dealt to several replicas (own handle + HIP stream, one weight blob) give bit-identical audio: same kernels, same
    shapes, only the stream differs."""
    hp = H.default_v23()
    m = _model(hp)
    if mode != "fp32":
        m.set_generator_dtype(torch.bfloat16)
        m.set_flow_dtype(torch.float16)
    lengths = [17, 24, 9, 22, 40, 20, 33, 12]
    utts = _utts(lengths)
    g = torch.Generator().manual_seed(6)
    noise = [(torch.randn(2, T, generator=g), torch.randn(hp.inter_channels, 16 * T, generator=g)) for T in lengths]
    kw = dict(sdp_ratio=0.5, noise_scale=0.6, noise_scale_w=0.9, length_scale=1.0, max_batch=2, max_pad_ratio=1.5)
    one = serving.synthesize(m, utts, noise=noise, **kw)
    for n in (2, 3):
        many = serving.synthesize(m, utts, noise=noise, requests_in_flight=n, **kw)
        assert len(serving.replicas(m, n)) == n and serving.replicas(m, n)[1][0]._blob is m._blob
        for a, b in zip(one, many):
            assert a.shape == b.shape and np.array_equal(a, b)
    pcm = serving.synthesize(m, utts, noise=noise, requests_in_flight=2, as_pcm16=True, **kw)
    ref = serving.synthesize(m, utts, noise=noise, as_pcm16=True, **kw)
    assert all(np.array_equal(a, b) and a.dtype == np.int16 for a, b in zip(pcm, ref))
