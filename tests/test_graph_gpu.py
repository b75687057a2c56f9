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
    m.enable_graphs(True)
    first = _run(m, batch, nw, nz, kw)            # captures both phases, then replays
    assert len(m._graphs) == 2 and all(m._lib.bv2_graph_num_nodes(g["graph"]) > 50 for g in m._graphs.values())
    again = _run(m, batch, nw, nz, kw)            # pure replay
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
    m.enable_graphs(True, static_io=True)
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
    m.enable_graphs(True, static_io=True)
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
