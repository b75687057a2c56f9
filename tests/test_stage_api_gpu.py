"""GPU: the six ONNX-seam stage calls (include/bv2.h "single stages", bert_vits2_amd/onnx_api.py) against the oracle's taps, with
the tensor names / layouts of the reference's exported graphs (onnx_modules/V230/models_onnx.py:896-1063), and the consumer
(`StageSession`, the counterpart of onnx_modules/V230_OnnxInference/__init__.py:36-126) end to end."""
import numpy as np
import pytest
import torch

from bert_vits2_amd import hparams as H, onnx_api
from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict, rms

pytestmark = pytest.mark.gpu
_M = {}


def _model(hp, seed):
    from bert_vits2_amd import models
    key = (hp.use_transformer_flow, seed)
    if key not in _M:
        m = models.from_hparams(hp)
        m.load_state_dict(cached_state_dict(hp, seed), strict=False)
        _M[key] = m.to("cuda").eval()
    return _M[key]


def maxrel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.mark.parametrize("name", ["mix_b2_ragged", "mid_b2_t72", "t3_b1"])
def test_each_stage_against_the_oracle(name):
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    m = _model(hp, seed)
    g = m.stage_emb_g(batch["sid"])
    assert torch.equal(g.cpu(), ref["g"][:, :, 0])                                  # a lookup: exact
    xout, m_p, logs_p, x_mask = m.stage_enc_p(batch["x"], batch["tone"], batch["language"], batch["bert"], batch["ja_bert"],
                                              batch["en_bert"], g, x_lengths=batch["x_lengths"])
    assert torch.equal(x_mask.cpu(), ref["x_mask"]) and x_mask.shape == ref["x_mask"].shape
    for t, k in ((xout, "enc_x"), (m_p, "enc_m"), (logs_p, "enc_logs")):
        assert maxrel(t, ref[k]) < 3e-4, k
    zin = nw * kw["noise_scale_w"]                                                  # the exported sdp graph takes SCALED noise
    logw_sdp = m.stage_sdp(ref["enc_x"], ref["x_mask"], zin, ref["g"])
    logw_dp = m.stage_dp(ref["enc_x"], ref["x_mask"], ref["g"])
    assert logw_sdp.shape == ref["logw_sdp"].shape
    assert (logw_sdp.cpu() - ref["logw_sdp"]).abs().max().item() < 1e-3
    assert (logw_dp.cpu() - ref["logw_dp"]).abs().max().item() < 1e-3
    z = m.stage_flow(ref["z_p"], None, ref["g"], y_mask=ref["y_mask"])              # the graph's own mask input
    assert maxrel(z, ref["z"]) < 1e-4
    z2 = m.stage_flow(ref["z_p"], ref["y_lengths"], ref["g"])
    assert torch.equal(z, z2)
    o = m.stage_generator(ref["z"] * ref["y_mask"], None, ref["g"])                 # dec.run({"z_in", "g"}): z_in as it is
    assert rms(o.cpu() - ref["o"]) < 2e-5
    with pytest.raises(ValueError):
        m.stage_flow(ref["z_p"], ref["y_lengths"], ref["g"], y_mask=ref["y_mask"])


def test_stage_session_consumer_matches_stagewise_oracle():
    """The reference consumer's call (numpy RNG seeded with `seed`, [T,1024] BERT layout, batch 1) through the HIP stages, against
    the oracle's stage functions glued the same way."""
    hp = H.default_v23()
    sd = cached_state_dict(hp, 0)
    m = _model(hp, 0)
    T = len(cases.ONNX_INFER_SYMBOLS)
    rng = np.random.RandomState(5)
    seq = np.array(cases.ONNX_INFER_SYMBOLS)
    tone, lang = np.zeros(T, np.int64), np.zeros(T, np.int64)
    berts = [rng.randn(T, 1024).astype(np.float32) for _ in range(3)]
    sid = np.array([3])
    kw = dict(seed=7, seq_noise_scale=0.6, sdp_noise_scale=0.9, sdp_ratio=0.5, length_scale=1.0)
    wav = onnx_api.StageSession(m)(seq, tone, lang, *berts, sid, **kw)
    # oracle: same draws (np.random.seed(seed); randn(B,2,T)*sdp_noise_scale; randn(B,C,T_y)), same stage order
    x = torch.from_numpy(seq)[None]
    b = [torch.from_numpy(v).t()[None] for v in berts]
    np.random.seed(kw["seed"])
    zin = torch.from_numpy(np.random.randn(1, 2, T)).float()                        # un-scaled; oracle scales by noise_scale_w
    g = torch.nn.functional.embedding(torch.from_numpy(sid), sd["emb_g.weight"])[:, :, None]
    h, mp, lp, xm = O.text_encoder(sd, hp, x, torch.tensor([T]), torch.from_numpy(tone)[None], torch.from_numpy(lang)[None], b[0], b[1], b[2], g)
    logw = O.sdp_reverse(sd, h, xm, g, zin, kw["sdp_noise_scale"]) * kw["sdp_ratio"] + O.duration_predictor(sd, h, xm, g) * (1 - kw["sdp_ratio"])
    w_ceil = torch.ceil(torch.exp(logw) * xm * kw["length_scale"])
    yl, ym, attn, me, le = O.length_regulate(w_ceil, xm, mp, lp)
    nz = torch.from_numpy(np.random.randn(1, hp.inter_channels, int(yl.max()))).float()
    z = O.flow_reverse(sd, hp, me + nz * torch.exp(le) * kw["seq_noise_scale"], ym, g)
    ref = O.generator(sd, hp, z * ym, g)
    assert wav.shape == tuple(ref.shape)
    assert rms(torch.from_numpy(wav) - ref) < 5e-5
    for s in onnx_api.STAGES:
        assert tuple(getattr(onnx_api.StageSession(m), s).get_inputs()) == onnx_api.INPUT_NAMES[s]
