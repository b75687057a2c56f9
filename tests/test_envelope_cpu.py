"""CPU: the accepted hyper-parameter envelope (round 6).

 * `bert_vits2_amd/hparams.ENVELOPE` and `csrc/bv2_model.cpp validate()` state the same ranges: every envelope golden case and every seeded random
   draw is accepted by bv2_create; one step outside any bound is rejected with a message naming the field.
 * When /root/reference is present (the build container): the fixture generator reproduces committed goldens BIT FOR BIT with several different
   models built in ONE process (round 5's generator cached reference nets under (flow variant, seed) and handed `rb2_b2_t14` the ResBlock1 net), and
   the oracle matches the LIVE reference on every random draw — so the oracle is pinned on the interior of the envelope, not only its corners."""
import dataclasses

import numpy as np
import pytest
import torch

from bert_vits2_amd import hparams as H, lib as L
from oracle import bv2_oracle as O, cases, ref_import
from tests.helpers import load_golden


def _create(hp):
    import ctypes as C
    lib = L.load()
    try:
        cfg = L.make_config(hp)
    except ValueError as e:                       # the Python mirror rejects a few combinations before the library sees them
        return -1, str(e)
    h = C.c_void_p()
    rc = lib.bv2_create(C.byref(cfg), C.byref(h))
    msg = "" if rc == 0 else lib.bv2_last_error(None).decode()
    if rc == 0:
        lib.bv2_destroy(h)
    return rc, msg


def test_every_envelope_case_and_random_draw_is_accepted():
    for name in cases.CASES:
        rc, msg = _create(cases.build_case(name)[0])
        assert rc == 0, (name, msg)
    for i in range(cases.N_RANDOM_HPARAMS):
        rc, msg = _create(cases.random_hparams(i)[0])
        assert rc == 0, (i, msg)


def test_random_draws_span_the_envelope():
    """The sampler must actually reach the bounds it claims (a range nobody draws from is an untested range)."""
    hps = [cases.random_hparams(i)[0] for i in range(cases.N_RANDOM_HPARAMS)] + [cases.build_case(n)[0] for n in cases.CASES]
    E = H.ENVELOPE
    seen = lambda f: {f(hp) for hp in hps}
    assert seen(lambda hp: hp.hidden_channels) == set(E["hidden_channels"])
    assert seen(lambda hp: hp.hidden_channels // hp.n_heads) == set(E["head_dim"])
    assert seen(lambda hp: hp.kernel_size) == set(E["kernel_size"])
    assert seen(lambda hp: hp.inter_channels) == set(E["inter_channels"])
    assert {min(seen(lambda hp: hp.n_flow_layer)), max(seen(lambda hp: hp.n_flow_layer))} == set(E["n_flow_layer"])
    assert {min(seen(lambda hp: hp.n_layers)), max(seen(lambda hp: hp.n_layers))} == set(E["n_layers"])
    assert {min(seen(lambda hp: len(hp.upsample_rates))), max(seen(lambda hp: len(hp.upsample_rates)))} == set(E["n_upsamples"])
    assert seen(lambda hp: len(hp.resblock_kernel_sizes)) == {1, 2, 3}
    assert {k for hp in hps for k in hp.resblock_kernel_sizes} == set(E["resblock_kernel"])
    assert {u for hp in hps for u in hp.upsample_rates} == set(E["upsample_rate"])
    assert {k // u for hp in hps for u, k in zip(hp.upsample_rates, hp.upsample_kernel_sizes)} == {1, 2, 3, 4}
    assert seen(lambda hp: hp.upsample_initial_channel >> len(hp.upsample_rates)) == set(E["final_generator_width"])
    assert {hp.use_transformer_flow for hp in hps} == {True, False} and {str(hp.resblock) for hp in hps} == {"1", "2"}
    for tf in (True, False):
        assert {hp.n_flow_layer % 2 for hp in hps if hp.use_transformer_flow == tf} == {0, 1}     # odd and even coupling counts on both flows


OUTSIDE = [
    # (overrides, word the message must contain)
    (dict(hidden_channels=64, n_heads=2), "hidden_channels"), (dict(hidden_channels=288, n_heads=3), "hidden_channels"),
    (dict(hidden_channels=144, n_heads=3), "hidden_channels"), (dict(hidden_channels=192, n_heads=4), "head dim"),
    (dict(hidden_channels=192, n_heads=1), "head dim"), (dict(hidden_channels=192, n_heads=5), "divisible"),
    (dict(filter_channels=64), "filter_channels"), (dict(filter_channels=1088), "filter_channels"), (dict(filter_channels=800), "filter_channels"),
    (dict(inter_channels=32), "inter_channels"), (dict(inter_channels=288), "inter_channels"), (dict(inter_channels=100), "inter_channels"),
    (dict(kernel_size=9), "kernel_size"), (dict(kernel_size=4), "kernel_size"),
    (dict(n_layers=2), "n_layers"), (dict(n_layers=9), "n_layers"), (dict(n_layers_trans_flow=2), "n_layers_trans_flow"),
    (dict(n_layers_trans_flow=9), "n_layers_trans_flow"), (dict(n_flow_layer=0), "n_flow_layer"), (dict(n_flow_layer=9), "n_flow_layer"),
    (dict(gin_channels=32), "gin_channels"), (dict(gin_channels=832), "gin_channels"), (dict(gin_channels=500), "gin_channels"),
    (dict(resblock_kernel_sizes=(3, 7, 13)), "resblock kernels"), (dict(resblock_kernel_sizes=(3, 7, 1)), "resblock kernels"),
    (dict(resblock_kernel_sizes=(3, 4, 7)), "resblock kernels"),
    (dict(resblock_dilation_sizes=((1, 3, 5), (1, 3, 13), (1, 3, 5))), "dilations"), (dict(resblock_dilation_sizes=((1, 3, 5), (0, 3, 5), (1, 3, 5))), "dilations"),
    (dict(upsample_rates=(8,), upsample_kernel_sizes=(16,), upsample_initial_channel=32), "upsampling stages"),
    (dict(upsample_rates=(2,) * 6, upsample_kernel_sizes=(4,) * 6, upsample_initial_channel=1024), "upsampling stages"),
    (dict(upsample_rates=(8, 8, 1, 2, 2), upsample_kernel_sizes=(16, 16, 1, 2, 2)), "rates"),
    (dict(upsample_rates=(8, 8, 2, 2, 2), upsample_kernel_sizes=(16, 16, 3, 2, 2)), "kernel"),
    (dict(upsample_rates=(8, 8, 2, 2, 2), upsample_kernel_sizes=(16, 16, 10, 2, 2)), "kernel"),
    (dict(upsample_rates=(8, 3, 2, 2, 2), upsample_kernel_sizes=(16, 6, 4, 2, 2)), "kernel"),          # (6 - 3) odd
    (dict(upsample_initial_channel=1024), "upsample_initial_channel"), (dict(upsample_initial_channel=256), "final Generator width"),
    (dict(upsample_initial_channel=512 - 32), "final Generator width"),
    (dict(upsample_rates=(8, 8), upsample_kernel_sizes=(16, 16), upsample_initial_channel=512), "final Generator width"),    # 128
]


@pytest.mark.parametrize("over,word", OUTSIDE)
def test_one_step_outside_the_envelope_is_rejected_with_a_message(over, word):
    rc, msg = _create(H.default_v23(**over))
    assert rc != 0, over
    assert word in msg, (over, msg)


def test_direct_c_callers_must_hand_over_exactly_the_dilations_the_reference_reads():
    """lib.make_config truncates longer dilation lists the way the reference ignores them; the C struct takes exactly 3 (ResBlock1) / 2 (ResBlock2)."""
    import ctypes as C
    lib = L.load()
    cfg = L.make_config(H.default_v23(resblock_dilation_sizes=((1, 3, 5, 7),) * 3))
    assert cfg.n_resblock_dilations == 3
    for rb, n, ok in ((1, 3, True), (1, 2, False), (1, 4, False), (2, 2, True), (2, 3, False)):
        cfg = L.make_config(H.default_v23())
        cfg.resblock_type, cfg.n_resblock_dilations = rb, n
        h = C.c_void_p()
        rc = lib.bv2_create(C.byref(cfg), C.byref(h))
        assert (rc == 0) == ok, (rb, n)
        if rc == 0:
            lib.bv2_destroy(h)
    with pytest.raises(ValueError):
        L.make_config(H.default_v23(resblock_dilation_sizes=((1, 3),) * 3))           # ResBlock1 would IndexError in the reference
    with pytest.warns(UserWarning):
        cfg = L.make_config(H.default_v23(resblock="3", resblock_dilation_sizes=((1, 3),) * 3))     # models.py:508: not "1" -> ResBlock2
    assert cfg.resblock_type == 2


needs_ref = pytest.mark.skipif(not ref_import.available(), reason="/root/reference not present (GPU box)")


@needs_ref
def test_generator_reproduces_committed_goldens_with_many_models_in_one_process():
    from oracle import gen_golden
    threads = torch.get_num_threads()
    gen_golden.deterministic()
    try:
        nets = {}
        # three different Generators / widths / flows under the SAME weight seed first (the round-5 cache-key bug), then envelope cases
        for name in ["zh_b1_t24", "rb2_b2_t14", "narrow_b2_t18", "wn_b1_t16", "hp03_tf2_h256x8_rb2", "hp09_wn5_h256x2"]:
            arrays, meta, _ = gen_golden.generate_case(name, nets)
            gmeta, gold = load_golden(name)
            assert set(arrays) == set(gold), name
            for k, v in arrays.items():
                assert np.array_equal(v, gold[k].numpy()), (name, k, float(np.abs(v - gold[k].numpy()).max()))
            assert meta["checksums"] == gmeta["checksums"] and meta["T_y"] == gmeta["T_y"]
        assert len(nets) == 6
    finally:
        torch.use_deterministic_algorithms(False)
        torch.set_num_threads(threads)


@needs_ref
@pytest.mark.parametrize("i", range(cases.N_RANDOM_HPARAMS))
def test_oracle_matches_the_live_reference_on_random_hparams(i):
    from bert_vits2_amd import synth
    hp, lens, langs, sids, seed = cases.random_hparams(i)
    sd = synth.synthetic_state_dict(hp, seed)
    batch = synth.synthetic_batch(lens, langs, sids)
    nw, nz = synth.synthetic_noise(len(lens), max(lens), 256, hp.inter_channels)
    net = ref_import.build_reference_net(hp, sd)
    ref = ref_import.reference_infer(net, batch, nw, nz, **cases.INFER_KW)
    out = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"], batch["ja_bert"],
                  batch["en_bert"], noise_w=nw, noise_z=nz, **cases.INFER_KW)
    assert torch.equal(out["w_ceil"], ref["w_ceil"]) and torch.equal(out["attn"], ref["attn"]) and torch.equal(out["y_mask"], ref["y_mask"])
    for k in ("z_p", "m_p", "logs_p", "z"):
        assert (out[k] - ref[k]).abs().max() <= 3e-4 * max(1.0, ref[k].abs().max().item()), k
    assert (out["o"] - ref["o"]).pow(2).mean().sqrt().item() <= 2e-5
