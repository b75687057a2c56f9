"""GPU: the bf16 Generator's stage hand-over (round 6).  The launch that finishes a stage's ResBlocks — the last pair launch of the wide stages
(respair_cl_bf16.hip, both forms), the whole-ResBlock launch at C = 16 (resblock_c16_bf16.hip) — runs the stage's n branches tile by tile in ONE
workgroup and writes one tensor, the branch mean of reference models.py:545-552 (`xs / self.num_kernels`); the next ConvTranspose1d / conv_post reads
one tensor instead of n.  With "stage_sum" = 0 the n branch tensors are handed over and the consumer forms the mean.  Both forms round at the same
points (kernels/cl_bf16.h stage_mean: branches summed widest kernel first, the running sum stored as bf16), so they must agree BIT FOR BIT — the
strongest check there is of the in-kernel branch loop, the common tile origin and the read-modify-write of the running sum — and both sit on the
oracle (oracle.generator_bf16, which restates those rounding points) within the bars of test_bf16_gpu.py."""
import pytest
import torch

from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict, load_golden, rms

pytestmark = pytest.mark.gpu


def _model(hp, seed):
    from bert_vits2_amd import models
    m = models.from_hparams(hp)
    m.load_state_dict(cached_state_dict(hp, seed), strict=False)
    m = m.to("cuda").eval()
    m.set_generator_dtype(torch.bfloat16)
    return m


def _bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _stage_mean(rs):
    """kernels/cl_bf16.h stage_mean on fp32 tensors holding bf16 values (branch 0 first in `rs`)."""
    if len(rs) == 1:
        return rs[0]
    s = rs[-1]
    for r in rs[-2:0:-1]:
        s = _bf(s + r)
    return _bf((s + rs[0]) * torch.tensor(1.0 / len(rs), dtype=torch.float32, device=s.device))


def _stage_shapes(hp, B, Ty):
    up, out = 1, []
    for i, u in enumerate(hp.upsample_rates):
        up *= u
        out.append((B, hp.upsample_initial_channel // 2 ** (i + 1), Ty * up))
    return out


# three ResBlock kernels (the released Generator), two, and a final width of 32 (conv_post on the any-width kernel, the last stage on the C = 32 pair kernel)
@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged", "narrow_b2_t18", "hp01_tf3_h128x4", "hp06_wn3_h128x2", "hp04_tf5_h192x6", "hp11_tf6_h192x2"])
def test_one_summed_tensor_equals_n_branch_tensors_bit_for_bit(name):
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    meta, gold = load_golden(name)
    sd = cached_state_dict(hp, seed)
    g = torch.nn.functional.embedding(batch["sid"], sd["emb_g.weight"])[:, :, None]
    m = _model(hp, seed)
    B, _, Ty = gold["z"].shape
    nk = len(hp.resblock_kernel_sizes)
    shapes = _stage_shapes(hp, B, Ty)
    # n branch tensors, with every branch tapped
    m.set_option("stage_sum", 0)
    rb = {(i, j): torch.zeros(*sh, device="cuda") for i, sh in enumerate(shapes) for j in range(nk)}
    for (i, j), t in rb.items():
        m.set_tap(f"dec.rb.{i}.{j}", t)
    o_n = m.stage_generator(gold["z"], gold["y_lengths"], g)
    torch.cuda.synchronize()
    m.set_tap(None)
    # one summed tensor, the stage outputs tapped
    m.set_option("stage_sum", 1)
    st = {i: torch.full(sh, float("nan"), device="cuda") for i, sh in enumerate(shapes)}
    for i, t in st.items():
        m.set_tap(f"dec.stage.{i}", t)
    o_1 = m.stage_generator(gold["z"], gold["y_lengths"], g)
    torch.cuda.synchronize()
    m.set_tap(None)
    o_plain = m.stage_generator(gold["z"], gold["y_lengths"], g)           # no taps at all: the default path
    assert torch.equal(o_1, o_n) and torch.equal(o_plain, o_1)
    assert float(o_1.abs().max()) > 0 and torch.isfinite(o_1).all()
    summed = 0
    for i in range(len(shapes)):
        if bool(torch.isnan(st[i]).all()):         # tap never written: this stage's finishing kernel has no summed form (e.g. k = 11 with dilation 8 at
            continue                               # C = 16 is outside resblock_c16_bf16.hip's guard rows) and handed n tensors over — same results, checked above
        want = _stage_mean([rb[(i, j)] for j in range(nk)])
        assert torch.equal(st[i], want), (i, float((st[i] - want).abs().max()))
        summed += 1
    assert summed >= len(shapes) - 1, summed
    # and the oracle restates the same rounding points
    with torch.no_grad():
        o16 = O.generator_bf16(sd, hp, gold["z"] * gold["y_mask"], g)
    e = rms(o_1.cpu() - o16) / rms(o16)
    assert e < 1e-2, e


@pytest.mark.parametrize("exact", [False, True])
def test_hand_over_at_a_multi_tile_size_and_with_exact_lengths(exact):
    """Every kernel of the hand-over over many tiles per batch item (T_y = 150 frames -> 76 800 rows at C = 16, several tiles of every pair kernel),
    ragged lengths, with and without exact_lengths (per-item row limits: tiles past an utterance's end are skipped in BOTH forms)."""
    from bert_vits2_amd import hparams as H
    hp = H.default_v23()
    m = _model(hp, 0)
    B, Ty = 3, 150
    gen = torch.Generator().manual_seed(3)
    z = torch.randn(B, hp.inter_channels, Ty, generator=gen)
    yl = torch.tensor([150, 97, 31])
    sd = cached_state_dict(hp, 0)
    g = torch.nn.functional.embedding(torch.tensor([1, 2, 3]), sd["emb_g.weight"])[:, :, None]
    outs = {}
    for v in (1, 0):
        m.set_option("stage_sum", v)
        outs[v] = m.stage_generator(z, yl, g)
        torch.cuda.synchronize()
    m.set_option("stage_sum", 1)
    assert torch.equal(outs[1], outs[0])
    assert torch.isfinite(outs[1]).all() and float(outs[1].abs().max()) > 0
    if exact:
        # infer() with exact_lengths on a ragged batch, both forms
        name = "mix_b2_ragged"
        hp2, seed, batch, nw, nz, kw = cases.build_case(name)
        args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
        res = {}
        for v in (1, 0):
            m.set_option("stage_sum", v)
            res[v] = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), exact_lengths=True, **kw)
        m.set_option("stage_sum", 1)
        ym = res[1][2]
        n = (ym.sum((1, 2)).long() * hp2.total_upsample).tolist()
        for b_, nb_ in enumerate(n):
            assert torch.equal(res[1][0][b_, 0, :nb_], res[0][0][b_, 0, :nb_]), b_


def test_hand_over_under_graph_replay():
    hp, seed, batch, nw, nz, kw = cases.build_case("mix_b2_ragged")
    m = _model(hp, seed)
    args = [batch[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert")]
    eager = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), **kw)[0].clone()
    m.enable_graphs(True, ty_bucket=1)
    try:
        for _ in range(2):
            o = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), **kw)[0]
        assert torch.equal(o, eager)
    finally:
        m.enable_graphs(False)
