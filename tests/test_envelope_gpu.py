"""GPU: the hyper-parameter envelope bv2_create() accepts, pinned against the REAL reference (round 6).

`oracle/cases.ENVELOPE` holds twelve models off the released config — both flow variants x odd / even coupling counts, hidden 128 / 192 / 256
with 2-8 heads (head dims 32 / 64 / 96 / 128), text-encoder kernel 3 / 5, 3-6 Encoder layers, gin 256 / 512, one to three ResBlock kernels out
of {3,5,7,9,11} with their own dilations, ResBlock1 / ResBlock2, three to five upsampling stages, inter != hidden — whose goldens the real
reference produced in one `python -m oracle.gen_golden` run.  fp32 end to end against those goldens is `test_parity_gpu.py` (it runs every
case of `cases.CASES`); here: the reduced-precision forms (bf16 Generator, fp16 flow) against the oracles that define their rounding points,
hipGraph replay, and the same models at a batch large enough for the LDS-tiled (not split-K) kernels, against the oracle.
Bars as in test_bf16_gpu.py / test_f16_flow_gpu.py / test_graph_gpu.py."""
import pytest
import torch

from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict, load_golden, rms, valid_wave_mask

pytestmark = pytest.mark.gpu

ENV = list(cases.ENVELOPE)
_MODELS = {}


def _model(name):
    from bert_vits2_amd import models
    if name not in _MODELS:
        hp, seed, *_ = cases.build_case(name)
        m = models.from_hparams(hp)
        m.load_state_dict(cached_state_dict(hp, seed), strict=False)
        _MODELS.clear()                               # one resident model at a time
        _MODELS[name] = m.to("cuda").eval()
    return _MODELS[name]


def _relrms(a, b):
    return rms(a - b) / max(rms(b), 1e-30)


def _args(batch):
    return [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]


@pytest.mark.parametrize("name", ENV)
def test_envelope_bf16_generator_and_f16_flow_vs_oracles(name):
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    meta, gold = load_golden(name)
    sd = cached_state_dict(hp, seed)
    g = torch.nn.functional.embedding(batch["sid"], sd["emb_g.weight"])[:, :, None]
    ym = gold["y_mask"]
    zin = gold["z"] * ym
    with torch.no_grad():
        o16 = O.generator_bf16(sd, hp, zin, g)
        z16 = O.flow_reverse(sd, hp, gold["z_p"], ym, g, None, "fp16")
    m = _model(name)
    yl = gold["y_lengths"]
    try:
        m.set_generator_dtype(torch.bfloat16)
        o = m.stage_generator(gold["z"], yl, g)
        torch.cuda.synchronize()
        e16, e32 = _relrms(o.cpu(), o16), _relrms(o.cpu(), gold["o"])
        assert torch.isfinite(o).all()
        assert e16 < 1e-2, ("bf16 Generator vs bf16 oracle", e16)
        assert e32 < 5e-2, ("bf16 Generator vs the reference's fp32 waveform", e32)
        m.set_flow_dtype(torch.float16)
        z = m.stage_flow(gold["z_p"], yl, g)
        torch.cuda.synchronize()
        f16, f32 = _relrms(z.cpu() * ym, z16 * ym), _relrms(z.cpu() * ym, gold["z"] * ym)
        assert torch.isfinite(z).all()
        assert f16 < 2e-3, ("fp16 flow vs fp16 oracle", f16)
        assert f32 < 5e-3, ("fp16 flow vs the reference's fp32 z", f32)
        # both switches on, end to end, durations pinned to the reference's: path exact, waveform at the bf16 level
        out = m.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), w_ceil=gold["w_ceil"], **kw)
        torch.cuda.synchronize()
        assert torch.equal(out[1].cpu(), gold["attn"])
        vm = valid_wave_mask(gold["y_lengths"], hp.total_upsample, out[0].shape[2]).expand_as(gold["o"])
        rel = rms((out[0].cpu() - gold["o"])[vm]) / rms(gold["o"][vm])
        assert rel < 5e-2, ("fp16 flow + bf16 Generator end to end", rel)
        print(f"\n[{name}] bf16 dec rel RMS {e16:.2e} (oracle) / {e32:.2e} (reference); fp16 flow {f16:.2e} / {f32:.2e}; both, end to end {rel:.2e}")
    finally:
        m.set_generator_dtype(torch.float32)
        m.set_flow_dtype(torch.float32)
    # the fp32 path is untouched by the switches
    of = m.stage_generator(gold["z"], yl, g)
    assert rms(of.cpu() - gold["o"]) < 5e-5


def _run(m, batch, nw, nz, kw):
    o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*_args(batch), noise_w=nw.cuda(), noise_z=nz.cuda(), **kw)
    torch.cuda.synchronize()
    return dict(o=o, attn=attn, y_mask=y_mask, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p)


@pytest.mark.parametrize("name", ["hp04_tf5_h192x6", "hp07_wn4_h256x4_rb2", "hp01_tf3_h128x4"])
@pytest.mark.parametrize("reduced", [False, True])
def test_envelope_graph_replay_is_bit_identical_to_eager(name, reduced):
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    m = _model(name)
    try:
        if reduced:
            m.set_generator_dtype(torch.bfloat16)
            m.set_flow_dtype(torch.float16)
        eager = _run(m, batch, nw, nz, kw)
        m.enable_graphs(True, ty_bucket=1)
        first = _run(m, batch, nw, nz, kw)
        again = _run(m, batch, nw, nz, kw)
        assert len(m._graphs) == 2
        for k, v in eager.items():
            assert torch.equal(first[k], v), k
            assert torch.equal(again[k], v), k
    finally:
        m.enable_graphs(False)
        m.set_generator_dtype(torch.float32)
        m.set_flow_dtype(torch.float32)


# the same models where B*T and B*T_y are past the split-K regime (every encoder / flow convolution on the LDS-tiled kernels, attention and
# LayerNorm with B > 1, the Generator's wide tiles): no golden (a fixture of that size is MBs) — the oracle, which the goldens above pin for
# exactly these hyper-parameters, is the checker
# (four of the twelve: hidden 128 / 192 / 256, both flows, odd and even coupling counts, ResBlock1 / 2, 3-5 stages — the GPU suite has a time limit)
@pytest.mark.parametrize("name", ["hp03_tf2_h256x8_rb2", "hp04_tf5_h192x6", "hp06_wn3_h128x2", "hp09_wn5_h256x2"])
def test_envelope_at_a_tiled_batch_vs_oracle(name):
    from bert_vits2_amd import synth
    hp, seed, *_ = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    lens = [40, 33, 37, 21, 40, 8, 29, 36, 17, 40]
    batch = synth.synthetic_batch(lens, languages=[i % 3 for i in range(len(lens))], sids=[(7 * i + 1) % hp.n_speakers for i in range(len(lens))])
    nw, nz = synth.synthetic_noise(len(lens), 40, 512, hp.inter_channels)
    kw = dict(cases.INFER_KW)
    with torch.no_grad():
        ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"], batch["ja_bert"],
                      batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    m = _model(name)
    o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), **kw)
    torch.cuda.synchronize()
    wc = m.last_encode["w_ceil"].cpu()[:, None]
    flips = (wc != ref["w_ceil"]).float().mean().item()
    assert flips <= 0.01, flips
    if flips > 0:
        o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)
        torch.cuda.synchronize()
    assert torch.equal(attn.cpu(), ref["attn"]) and torch.equal(y_mask.cpu(), ref["y_mask"])
    for t, k in ((z_p, "z_p"), (m_p, "m_p"), (logs_p, "logs_p"), (z, "z")):
        d = ((t.cpu() - ref[k]).abs().max() / ref[k].abs().max()).item()
        assert d < 3e-4, (k, d)
    vm = valid_wave_mask(ref["y_lengths"], hp.total_upsample, o.shape[2]).expand_as(ref["o"])
    err = rms((o.cpu() - ref["o"])[vm])
    assert err <= 5e-5, err
    # reduced precision at this batch: the tiled fp16 Encoder / WN kernels and the wide bf16 Generator tiles
    try:
        m.set_generator_dtype(torch.bfloat16)
        m.set_flow_dtype(torch.float16)
        with torch.no_grad():
            z16 = O.flow_reverse(sd, hp, ref["z_p"], ref["y_mask"], ref["g"], None, "fp16")
            o16 = O.generator_bf16(sd, hp, ref["z"] * ref["y_mask"], ref["g"])
        zh = m.stage_flow(ref["z_p"], ref["y_lengths"], ref["g"])
        oh = m.stage_generator(ref["z"], ref["y_lengths"], ref["g"])
        torch.cuda.synchronize()
        ym = ref["y_mask"]
        f16, e16 = _relrms(zh.cpu() * ym, z16 * ym), _relrms(oh.cpu()[vm], o16[vm])
        print(f"\n[{name}] B={len(lens)} T_y={ref['y_mask'].shape[2]}: fp32 wave RMS err {err:.2e}; fp16 flow {f16:.2e}, bf16 dec {e16:.2e} (rel, vs their oracles)")
        assert f16 < 2e-3, f16
        assert e16 < 1e-2, e16
    finally:
        m.set_generator_dtype(torch.float32)
        m.set_flow_dtype(torch.float32)


# the interior of the envelope: seeded random models (cases.random_hparams draws from hparams.ENVELOPE = validate()'s ranges) against the oracle,
# which tests/test_envelope_cpu.py holds to the LIVE reference on these very draws in the build container
# (every other draw on the GPU — the suite has a time limit and each draw costs a model build plus three CPU oracle runs; tests/test_envelope_cpu.py holds
# the oracle to the live reference on ALL draws, and the twelve ENVELOPE models above cover the corners)
@pytest.mark.parametrize("i", range(0, cases.N_RANDOM_HPARAMS, 2))
def test_random_hparams_vs_oracle(i):
    from bert_vits2_amd import models, synth
    hp, lens, langs, sids, seed = cases.random_hparams(i)
    sd = synth.synthetic_state_dict(hp, seed)
    batch = synth.synthetic_batch(lens, langs, sids)
    nw, nz = synth.synthetic_noise(len(lens), max(lens), 256, hp.inter_channels)
    kw = dict(cases.INFER_KW)
    with torch.no_grad():
        ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"], batch["ja_bert"],
                      batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    _MODELS.clear()
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), **kw)
    torch.cuda.synchronize()
    wc = m.last_encode["w_ceil"].cpu()[:, None]
    flips = (wc != ref["w_ceil"]).float().mean().item()
    assert flips <= 0.05, flips                       # T is 3..12 symbols: one flip is already 3-8 %
    if flips > 0:
        o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)
        torch.cuda.synchronize()
    assert torch.equal(attn.cpu(), ref["attn"]) and torch.equal(y_mask.cpu(), ref["y_mask"])
    for t, k in ((z_p, "z_p"), (m_p, "m_p"), (logs_p, "logs_p"), (z, "z")):
        d = ((t.cpu() - ref[k]).abs().max() / ref[k].abs().max()).item()
        assert d < 3e-4, (k, d)
    vm = valid_wave_mask(ref["y_lengths"], hp.total_upsample, o.shape[2]).expand_as(ref["o"])
    err = rms((o.cpu() - ref["o"])[vm])
    assert err <= 5e-5, err
    m.set_generator_dtype(torch.bfloat16)
    m.set_flow_dtype(torch.float16)
    with torch.no_grad():
        z16 = O.flow_reverse(sd, hp, ref["z_p"], ref["y_mask"], ref["g"], None, "fp16")
        o16 = O.generator_bf16(sd, hp, ref["z"] * ref["y_mask"], ref["g"])
    zh = m.stage_flow(ref["z_p"], ref["y_lengths"], ref["g"])
    oh = m.stage_generator(ref["z"], ref["y_lengths"], ref["g"])
    torch.cuda.synchronize()
    ym = ref["y_mask"]
    f16, e16 = _relrms(zh.cpu() * ym, z16 * ym), _relrms(oh.cpu()[vm], o16[vm])
    print(f"\n[random {i}] fp32 wave RMS err {err:.2e} (flips {flips:.3f}); fp16 flow {f16:.2e}, bf16 dec {e16:.2e} (rel, vs their oracles)")
    assert f16 < 2e-3, f16
    assert e16 < 1e-2, e16
