"""GPU: hipGraph replay of the two phases (bv2_graph_capture_encode / _decode, BASELINE config 3 "hipGraph-captured
decode") must be BIT-IDENTICAL to the eager launch sequence it recorded — same kernels, same arguments, same order —
also when the graph is replayed with new inputs of the same shape, and after a dtype switch forces a re-capture."""
import pytest
import torch

from oracle import cases
from tests.helpers import cached_state_dict

pytestmark = pytest.mark.gpu


def _model(hp, seed):
    from bert_vits2_amd import models
    m = models.from_hparams(hp)
    m.load_state_dict(cached_state_dict(hp, seed), strict=False)
    return m.to("cuda").eval()


def _run(m, batch, nw, nz, kw):
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*args, noise_w=nw.cuda(), noise_z=nz.cuda(), **kw)
    torch.cuda.synchronize()
    return dict(o=o, attn=attn, y_mask=y_mask, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p)


@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged"])
def test_graph_replay_is_bit_identical_to_eager(name):
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    m = _model(hp, seed)
    eager = _run(m, batch, nw, nz, kw)
    # a second input set of the same shape (other features / noise): the replayed graph must follow the data
    g = torch.Generator().manual_seed(7)
    batch2 = dict(batch)
    for k in ("bert", "ja_bert", "en_bert"):
        batch2[k] = torch.randn(batch[k].shape, generator=g)
    nw2, nz2 = torch.randn(nw.shape, generator=g), torch.randn(nz.shape, generator=g)
    eager2 = _run(m, batch2, nw2, nz2, kw)
    m.enable_graphs(True, ty_bucket=1)        # exact T_y keys: replay == the eager launch sequence, bit for bit
    first = _run(m, batch, nw, nz, kw)            # captures both phases, then replays
    assert len(m._graphs) == 2 and all(m._lib.bv2_graph_num_nodes(g["graph"]) > 50 for g in m._graphs.values())
    again = _run(m, batch, nw, nz, kw)            # pure replay
    # ... and again on the SAME inputs: state a replay leaves behind (the x3 convs' max |x| slots, conv_x6.hip) must not leak into the next
    for _ in range(2):
        rep = _run(m, batch, nw, nz, kw)
        assert torch.equal(rep["o"], eager["o"])
    other = _run(m, batch2, nw2, nz2, kw)         # replay with new data (same shapes only if the durations agree)
    for k, v in eager.items():
        assert torch.equal(first[k], v), k
        assert torch.equal(again[k], v), k
    for k, v in eager2.items():
        assert other[k].shape == v.shape and torch.equal(other[k], v), k
    # a dtype switch drops the recorded graphs; the re-captured ones run the new arithmetic
    m.set_generator_dtype(torch.bfloat16)
    assert len(m._graphs) == 0
    gb = _run(m, batch, nw, nz, kw)
    m.enable_graphs(False)
    eb = _run(m, batch, nw, nz, kw)
    assert torch.equal(gb["o"], eb["o"]) and not torch.equal(gb["o"], eager["o"])


def test_capture_refuses_default_stream_and_taps():
    hp, seed, batch, nw, nz, kw = cases.build_case("zh_b1_t24")
    m = _model(hp, seed)
    m.enable_graphs(True)
    t = torch.zeros(1, hp.hidden_channels, batch["x"].shape[1], device="cuda")
    m.set_tap("enc.x0", t)                        # with a tap set the shim stays on the eager path (taps are debug only)
    _run(m, batch, nw, nz, kw)
    assert len(m._graphs) == 0 and float(t.abs().sum()) > 0
    m.set_tap(None)


def test_static_io_replay_reads_inputs_in_place_and_matches_eager():
    """enable_graphs(static_io=True): no staging copies — the graph reads the caller's tensors in place and hands out its own output
    buffers; the device draw of models.py:1071 is made in place with the reference's strides (seeded run == seeded eager run)."""
    hp, seed, batch, nw, nz, kw = cases.build_case("mix_b2_ragged")
    m = _model(hp, seed)
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    torch.manual_seed(11)
    eager = m.infer(*args, **kw)
    e_o, e_z = eager[0].clone(), eager[3][0].clone()
    m.enable_graphs(True, static_io=True, ty_bucket=1)
    torch.manual_seed(11)
    o1 = m.infer(*args, **kw)[0]
    assert torch.equal(o1, e_o)
    n_graphs = len(m._graphs)
    torch.manual_seed(11)
    o2, _, _, (z2, *_r) = m.infer(*args, **kw)            # pure replay on the same tensors: no new recording
    assert len(m._graphs) == n_graphs and torch.equal(o2, e_o) and torch.equal(z2, e_z)
    assert o2.data_ptr() == o1.data_ptr()                 # the graph's own output buffer
    # new CONTENT in the same input tensors is followed (read in place) ...
    args[5].mul_(0.5)
    torch.manual_seed(11)
    o3 = m.infer(*args, **kw)[0].clone()
    m.enable_graphs(False)
    torch.manual_seed(11)
    assert torch.equal(m.infer(*args, **kw)[0], o3) and not torch.equal(o3, e_o)
    # ... and a tensor at ANOTHER address never reads stale memory: the shape's graph is re-recorded ONCE with input buffers of its
    # own (inputs copied in from then on), so a caller that re-materialises its inputs every call does not re-capture every call
    m.enable_graphs(True, static_io=True, ty_bucket=1)
    torch.manual_seed(11)
    m.infer(*args, **kw)
    n1 = len(m._graphs)
    assert all(not e["own_inputs"] for e in m._graphs.values())
    args2 = [a.clone() for a in args]
    torch.manual_seed(11)
    o4 = m.infer(*args2, **kw)[0].clone()
    assert len(m._graphs) == n1 and torch.equal(o4, o3)
    entry_a = next(e for k, e in m._graphs.items() if k[0] == "A")
    assert entry_a["own_inputs"]
    handle = entry_a["graph"].value
    args3 = [a.clone() for a in args]
    args3[5].mul_(2.0)                                    # the original content again, at a third address
    torch.manual_seed(11)
    o5 = m.infer(*args3, **kw)[0]
    assert next(e for k, e in m._graphs.items() if k[0] == "A")["graph"].value == handle      # replayed, not re-recorded
    assert torch.equal(o5, e_o)


# ---- T_y buckets (round 6): T_y = max(y_lengths) is data-dependent (reference commons.py:119-123, models.py:1058), so phase B is recorded once per
# 32-frame bucket and replayed for every T_y inside it
def _relrms(a, b):
    return float(((a - b).double().pow(2).mean().sqrt()) / b.double().pow(2).mean().sqrt().clamp_min(1e-30))


@pytest.mark.parametrize("name,exact", [("mix_b2_ragged", False), ("mix_b2_ragged", True), ("zh_b1_t24", False), ("wn_b2_t40", False)])
def test_ty_bucket_replay_serves_every_ty_of_a_bucket(name, exact):
    """Eight requests of one shape whose durations (hence T_y) differ: eager at the exact T_y (the reference's semantics) first, then graphs with the
    default 32-frame bucket.  Captures = 1 (phase A) + the number of distinct buckets; every replayed result has the eager result's shapes, the same
    integer outputs, and the same floats up to fp32 summation order (a bucket may pick another split-K factor than the exact T_y)."""
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    m = _model(hp, seed)
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    g = torch.Generator().manual_seed(5)
    noises = [(nw, nz)] + [(torch.randn(nw.shape, generator=g) * 1.5, torch.randn(nz.shape, generator=g)) for _ in range(7)]
    run = lambda w, z, **k: m.infer(*args, noise_w=w, noise_z=z.cuda(), exact_lengths=exact, **kw, **k)
    eager = [run(w, z) for w, z in noises]
    tys = [int(e[2].shape[2]) for e in eager]
    buckets = {(t + 31) // 32 for t in tys}
    assert len(set(tys)) >= 3, tys                                    # the case really produces several T_y
    eager_b = [run(w, z, ty_bucket=32) for w, z in noises]            # the eager launch sequence AT the bucket: what the graph records
    m.enable_graphs(True)                                             # default bucket: 32 frames
    try:
        got = [run(w, z) for w, z in noises]
        torch.cuda.synchronize()
        assert m.graph_stats["captures"] == 1 + len(buckets), (m.graph_stats, tys)
        assert m.graph_stats["replays"] == 2 * len(noises) - m.graph_stats["captures"]
        for e, eb, r, ty in zip(eager, eager_b, got, tys):
            o, attn, ym, (z, z_p, m_p, logs_p) = r
            assert o.shape == e[0].shape and attn.shape == e[1].shape and ym.shape == e[2].shape and z.shape == e[3][0].shape
            assert torch.equal(attn, e[1]) and torch.equal(ym, e[2])
            assert torch.equal(m_p, e[3][2]) and torch.equal(logs_p, e[3][3]) and torch.equal(z_p, e[3][1])       # gathers: no arithmetic to reorder
            assert torch.equal(o, eb[0]) and torch.equal(z, eb[3][0])                                              # replay == eager at the bucket
            valid = slice(0, int(ym.sum((1, 2)).min().item()) * hp.total_upsample) if exact else slice(None)
            assert _relrms(z * ym, e[3][0] * ym) < 1e-5
            assert _relrms(o[..., valid], e[0][..., valid]) < 1e-4, (ty, _relrms(o[..., valid], e[0][..., valid]))
            assert torch.isfinite(o).all()
    finally:
        m.enable_graphs(False)


def test_ty_bucket_keeps_the_seeded_rng_contract_and_reduced_precision():
    """No injected noise: the device draw of models.py:1071 stays `randn_like(m_p)` over exactly [B, C, T_y] in the reference's memory order also when
    the graph's noise buffer is a bucket wide (seeded graph run == seeded eager run up to summation order); bf16 Generator + fp16 flow at a bucket."""
    hp, seed, batch, nw, nz, kw = cases.build_case("mix_b2_ragged")
    m = _model(hp, seed)
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    torch.manual_seed(11)
    e = m.infer(*args, **kw)
    assert e[2].shape[2] % 32 != 0
    try:
        m.enable_graphs(True)
        torch.manual_seed(11)
        r = m.infer(*args, **kw)
        assert torch.equal(r[3][1], e[3][1])                          # z_p = m_p + noise * exp(logs_p) * scale: the same noise, element for element
        assert r[0].shape == e[0].shape and _relrms(r[0], e[0]) < 1e-4
        m.enable_graphs(False)
        m.set_generator_dtype(torch.bfloat16)
        m.set_flow_dtype(torch.float16)
        eh = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), **kw)
        ehb = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), ty_bucket=32, **kw)
        m.enable_graphs(True)
        rh = [m.infer(*args, noise_w=nw, noise_z=nz.cuda(), **kw) for _ in range(2)][1]
        assert torch.equal(rh[0], ehb[0]) and rh[0].shape == eh[0].shape
        assert _relrms(rh[0], eh[0]) < 2e-2                           # bf16 level: another tile split moves a few bf16 roundings
    finally:
        m.enable_graphs(False)
        m.set_generator_dtype(torch.float32)
        m.set_flow_dtype(torch.float32)


def test_ty_bucket_with_max_len_and_without_attn():
    """`max_len` (reference models.py:1073: dec sees z[:, :, :max_len]) cuts the Generator below T_y: under a bucket the flow still runs bucket-wide while
    the Generator keeps the exact cut; `want_attn=False` (serving) leaves the path out of the outputs in both forms."""
    hp, seed, batch, nw, nz, kw = cases.build_case("mix_b2_ragged")
    m = _model(hp, seed)
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    full = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), **kw)
    Ty = full[2].shape[2]
    ml = Ty - 7
    e = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), max_len=ml, want_attn=False, **kw)
    assert e[0].shape[2] == ml * hp.total_upsample and e[1] is None
    assert torch.equal(e[0][..., : (ml - 16) * hp.total_upsample], full[0][..., : (ml - 16) * hp.total_upsample]) or \
        _relrms(e[0][..., : (ml - 16) * hp.total_upsample], full[0][..., : (ml - 16) * hp.total_upsample]) < 1e-4
    m.enable_graphs(True)
    try:
        for _ in range(2):
            r = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), max_len=ml, want_attn=False, **kw)
        assert r[1] is None and r[0].shape == e[0].shape and r[2].shape == e[2].shape and r[3][0].shape == e[3][0].shape
        assert _relrms(r[0], e[0]) < 1e-4 and torch.equal(r[2], e[2])
    finally:
        m.enable_graphs(False)
