"""Parity cases shared by ``gen_golden.py`` (which runs the REAL reference on them), the CPU tests
(oracle vs golden) and the GPU tests (HIP vs oracle, HIP vs golden).  Test infrastructure only.

Everything is regenerated from seeds (weights: ``synth.synthetic_state_dict``; utterances:
``synth.synthetic_batch``; noise: ``synth.synthetic_noise``) so a fixture only has to carry the reference's
OUTPUTS plus a few weight checksums that prove both sides rebuilt the same checkpoint.
"""
from __future__ import annotations

import torch

from bert_vits2_amd import hparams as H, synth

# the fixed symbol sequence of reference onnx_infer.py:17-49 — the closest thing the reference has to a fixture
ONNX_INFER_SYMBOLS = [0, 97, 0, 8, 0, 78, 0, 8, 0, 76, 0, 37, 0, 40, 0, 97, 0, 8, 0, 23, 0, 8, 0, 74, 0, 26, 0, 104, 0]

INFER_KW = dict(noise_scale=0.6, noise_scale_w=0.9, sdp_ratio=0.5, length_scale=1.0)   # webui defaults, webui.py:443-454

CASES = {
    # name: hp overrides, lengths, languages, sids, weight seed, infer kwargs
    "zh_b1_t24": dict(hp={}, lengths=[24], languages=[0], sids=[0], seed=0, kw=INFER_KW),
    "mix_b2_ragged": dict(hp={}, lengths=[20, 13], languages=[1, 2], sids=[5, 700], seed=0, kw=INFER_KW),
    "onnx_fixture": dict(hp={}, lengths=[len(ONNX_INFER_SYMBOLS)], languages=[0], sids=[0], seed=0,
                         kw=dict(noise_scale=0.667, noise_scale_w=0.8, sdp_ratio=0.0, length_scale=1.0),
                         symbols=ONNX_INFER_SYMBOLS),
    "wn_b1_t16": dict(hp=dict(use_transformer_flow=False), lengths=[16], languages=[0], sids=[11], seed=0,
                      kw=dict(noise_scale=0.6, noise_scale_w=0.9, sdp_ratio=1.0, length_scale=1.2)),
    "short_b3": dict(hp={}, lengths=[3, 9, 1], languages=[0, 1, 2], sids=[1, 2, 3], seed=0, kw=INFER_KW),
    # padded T < window+1 = 5: the reference's relative-position helpers SLICE the embedding table instead of padding it
    # (attentions.py:345-357); T = 1 is the degenerate single-blank utterance
    "t1_b1": dict(hp={}, lengths=[1], languages=[0], sids=[7], seed=0, kw=INFER_KW),
    "t3_b1": dict(hp={}, lengths=[3], languages=[1], sids=[8], seed=0, kw=INFER_KW),
    "t4_b2": dict(hp={}, lengths=[4, 2], languages=[2, 0], sids=[9, 10], seed=0, kw=INFER_KW),
    # T >= 64 (several 32-column tiles per row in every kernel, T_y of a few hundred frames); also carries the reference's
    # own reduced-precision runs (dec under bf16 autocast, flow under fp16 autocast) — see AUTOCAST_KEYS
    "mid_b2_t72": dict(hp={}, lengths=[72, 64], languages=[0, 2], sids=[3, 421], seed=0, kw=INFER_KW, autocast=True),
    # also carries the reference's autocast runs: `flow` here is the ResidualCouplingBlock / WN stack under fp16 autocast
    "wn_b2_t40": dict(hp=dict(use_transformer_flow=False), lengths=[40, 33], languages=[0, 1], sids=[11, 12], seed=0,
                      kw=INFER_KW, autocast=True),
    # `resblock: "2"` (reference models.py:508, modules.py:318-363): ONE weight-normed conv per dilation, VITS's small-vocoder setting
    "rb2_b2_t14": dict(hp=dict(resblock="2", resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12))),
                       lengths=[14, 9], languages=[0, 2], sids=[4, 77], seed=0, kw=INFER_KW),
    # a NARROWER model than any released config — every width the kernels template on takes another value: hidden 128 (head dim 64), FFN 512,
    # inter 128 (the flow's half = 64), 4 + 3 Encoder layers, 3 couplings, gin 256, three upsampling stages 8 x 4 x 2 with kernels 16 / 8 / 4
    # (2 taps per phase everywhere; final Generator width 32)
    "narrow_b2_t18": dict(hp=dict(hidden_channels=128, filter_channels=512, inter_channels=128, n_layers=4, n_layers_trans_flow=3,
                                  n_flow_layer=3, gin_channels=256, upsample_rates=(8, 4, 2), upsample_kernel_sizes=(16, 8, 4),
                                  upsample_initial_channel=256),
                          lengths=[18, 11], languages=[0, 1], sids=[2, 640], seed=0, kw=INFER_KW),
}

# ---- the hyper-parameter envelope (round 6) --------------------------------------------------------------------------------------------
# Twelve models off the released config, each run by the REAL reference in ONE gen_golden invocation: both flow variants x odd and even
# coupling counts, hidden 128 / 192 / 256 with 2-8 heads (head dims 32 / 64 / 96 / 128), text-encoder kernel 3 / 5, 3-6 Encoder layers,
# gin 256 / 512, one to three ResBlock kernels from {3,5,7,9,11} with their own dilations, ResBlock1 and ResBlock2, three to five
# upsampling stages, inter != hidden.  What bv2_create's validate() accepts is what these pin (tests/test_envelope_*.py); everything
# else is rejected with a message.
def _hp(flow, n_flow, hidden, heads, filt, inter, ks, nl, nltf, gin, rb, rbk, rbd, ups, upk, init):
    return dict(use_transformer_flow=flow == "tf", n_flow_layer=n_flow, hidden_channels=hidden, n_heads=heads, filter_channels=filt,
                inter_channels=inter, kernel_size=ks, n_layers=nl, n_layers_trans_flow=nltf, gin_channels=gin, resblock=rb,
                resblock_kernel_sizes=rbk, resblock_dilation_sizes=rbd, upsample_rates=ups, upsample_kernel_sizes=upk,
                upsample_initial_channel=init)


_D135 = (1, 3, 5)
ENVELOPE = {
    "hp01_tf3_h128x4": dict(hp=_hp("tf", 3, 128, 4, 384, 64, 5, 3, 3, 256, "1", (3, 5), (_D135, (1, 2, 4)), (8, 4, 2), (16, 8, 4), 128),
                            lengths=[9, 6], languages=[0, 1], sids=[3, 40], seed=1),
    "hp02_tf4_h256x2": dict(hp=_hp("tf", 4, 256, 2, 512, 192, 3, 4, 5, 512, "1", (7,), (_D135,), (4, 4, 2, 2), (8, 8, 4, 4), 256),
                            lengths=[10], languages=[2], sids=[8], seed=2),
    "hp03_tf2_h256x8_rb2": dict(hp=_hp("tf", 2, 256, 8, 768, 128, 5, 5, 3, 256, "2", (3, 7, 11), ((1, 3), (2, 6), (3, 12)),
                                       (8, 8, 2, 2, 2), (16, 16, 4, 4, 4), 512),
                                lengths=[7, 8], languages=[0, 2], sids=[1, 849], seed=3),
    "hp04_tf5_h192x6": dict(hp=_hp("tf", 5, 192, 6, 640, 96, 3, 6, 6, 512, "1", (5, 9, 11), (_D135, _D135, (2, 4, 8)),
                                   (5, 5, 4), (15, 15, 8), 128),
                            lengths=[11, 5, 3], languages=[1, 0, 2], sids=[5, 6, 7], seed=4),
    "hp05_tf1_h192x3": dict(hp=_hp("tf", 1, 192, 3, 768, 192, 3, 3, 4, 256, "1", (3, 7, 11), (_D135, _D135, _D135),
                                   (8, 8, 2, 2), (16, 16, 4, 4), 256),
                            lengths=[8], languages=[0], sids=[0], seed=5),
    "hp06_wn3_h128x2": dict(hp=_hp("wn", 3, 128, 2, 512, 128, 3, 4, 4, 256, "1", (3, 9), (_D135, (1, 2, 3)), (8, 8, 4), (16, 16, 8), 128),
                            lengths=[9, 4], languages=[2, 1], sids=[10, 20], seed=6),
    "hp07_wn4_h256x4_rb2": dict(hp=_hp("wn", 4, 256, 4, 1024, 256, 5, 3, 4, 512, "2", (5,), ((1, 3),), (4, 4, 4, 2, 2),
                                       (8, 8, 8, 4, 4), 512),
                                lengths=[6, 7], languages=[0, 0], sids=[100, 200], seed=7),
    "hp08_wn1_h192x2": dict(hp=_hp("wn", 1, 192, 2, 768, 64, 3, 6, 4, 512, "1", (3, 5, 7), (_D135, _D135, _D135),
                                   (8, 8, 2, 2, 2), (16, 16, 4, 4, 4), 512),
                            lengths=[8], languages=[1], sids=[33], seed=8),
    "hp09_wn5_h256x2": dict(hp=_hp("wn", 5, 256, 2, 256, 96, 5, 5, 4, 256, "1", (11,), ((1, 3, 5),), (8, 8, 8), (16, 16, 16), 128),
                            lengths=[7, 9], languages=[2, 0], sids=[2, 3], seed=9),
    "hp10_wn2_h128x4_rb2": dict(hp=_hp("wn", 2, 128, 4, 128, 160, 3, 3, 4, 512, "2", (3, 5), ((1, 2), (3, 5)), (8, 4, 4, 2),
                                       (16, 8, 8, 4), 256),
                                lengths=[5, 10], languages=[1, 2], sids=[4, 5], seed=10),
    "hp11_tf6_h192x2": dict(hp=_hp("tf", 6, 192, 2, 384, 128, 5, 4, 3, 512, "1", (3, 7, 11), (_D135, _D135, _D135),
                                   (8, 8, 2, 2, 2), (16, 16, 4, 4, 4), 512),
                            lengths=[9], languages=[0], sids=[77], seed=11),
    "hp12_tf4_h128x2_rb2": dict(hp=_hp("tf", 4, 128, 2, 256, 256, 3, 5, 4, 256, "2", (7, 9, 11), ((1, 2), (2, 4), (3, 9)),
                                       (2, 2, 2, 2, 2), (4, 4, 4, 4, 4), 512),
                                lengths=[8, 5], languages=[0, 1], sids=[9, 19], seed=12),
}
for _n, _c in ENVELOPE.items():
    _c.setdefault("kw", INFER_KW)
CASES.update(ENVELOPE)
# cases whose fixture also stores the reference's autocast runs
AUTOCAST_CASES = [n for n, c in CASES.items() if c.get("autocast")] + ["mix_b2_ragged"]

T_Y_CAP = 1024     # noise_z is generated for this many frames and sliced to the realised T_y


def build_case(name: str):
    c = CASES[name]
    hp = H.default_v23(**c["hp"])
    batch = synth.synthetic_batch(c["lengths"], c["languages"], c["sids"])
    if "symbols" in c:
        s = torch.tensor(c["symbols"], dtype=torch.int64)
        batch["x"][0, : len(s)] = s
        batch["tone"][0] = 0
    B, T = batch["x"].shape
    noise_w, noise_z = synth.synthetic_noise(B, T, T_Y_CAP, hp.inter_channels)
    return hp, c["seed"], batch, noise_w, noise_z, dict(c["kw"])


CHECKSUM_KEYS = ["enc_p.emb.weight", "dec.resblocks.4.convs1.1.weight_v", "flow.flows.2.post.weight", "emb_g.weight"]


def weight_checksums(sd):
    return {k: float(sd[k].double().sum()) for k in CHECKSUM_KEYS if k in sd}


# reference run with `dec` under torch.autocast("cpu", bfloat16) on the fp32 z; `flow` under torch.autocast("cpu", float16)
# on the fp32 z_p (and the fp32 / bf16-autocast dec on that z): what BASELINE configs 3 / 5 mean on the reference side
AUTOCAST_KEYS = ["o_bf16dec", "z_f16flow", "o_f16flow", "o_f16flow_bf16dec"]

# un-injected run: torch.manual_seed(SEEDED_SEED); net.infer(...) — pins the RNG contract of SURVEY.md §8(b)
SEEDED_CASE = "mix_b2_ragged"
SEEDED_SEED = 20240923

GOLDEN_KEYS = ["o", "z", "z_p", "m_p", "logs_p", "enc_x", "enc_m", "enc_logs", "logw", "logw_sdp", "logw_dp", "w_ceil",
               "y_mask", "attn"]


# ---- seeded random models drawn from the WHOLE accepted envelope ------------------------------------------------------------------------
# hparams.ENVELOPE states the ranges bv2_create's validate() accepts (and rejects outside of, tests/test_envelope_cpu.py).  The twelve
# ENVELOPE cases above pin the oracle to the real reference across those ranges; these draws fill the interior on the GPU with the oracle as
# the checker (tests/test_envelope_gpu.py::test_random_hparams_vs_oracle).
N_RANDOM_HPARAMS = 24


def random_hparams(i: int):
    """(hp, lengths, languages, sids, weight seed) of the i-th seeded draw from hparams.ENVELOPE."""
    import random
    E = H.ENVELOPE
    r = random.Random(9000 + i)
    hidden = r.choice(E["hidden_channels"])
    heads = r.choice([h for h in range(1, 9) if hidden % h == 0 and (hidden // h) in E["head_dim"]])
    n_k = r.randint(*E["n_resblock_kernels"])
    rb = r.choice(["1", "2"])
    n_d = 3 if rb == "1" else 2
    while True:
        n_up = r.randint(*E["n_upsamples"])
        rates = [r.choice(E["upsample_rate"]) for _ in range(n_up)]
        total = 1
        for u in rates:
            total *= u
        if total <= 1024:
            break
    ks = []
    for u in rates:
        j = r.choice([j for j in range(1, E["upsample_taps_per_phase"][1] + 1) if ((j - 1) * u) % 2 == 0])
        ks.append(u * j)
    final = r.choice(E["final_generator_width"])
    init = final << n_up
    while init > E["upsample_initial_channel"][1]:
        final //= 2
        init = final << n_up
    if final < 16:
        return random_hparams(i + 1000)
    hp = H.default_v23(
        use_transformer_flow=r.random() < 0.5, n_flow_layer=r.randint(*E["n_flow_layer"]), hidden_channels=hidden, n_heads=heads,
        filter_channels=r.choice(E["filter_channels"]), inter_channels=r.choice(E["inter_channels"]), kernel_size=r.choice(E["kernel_size"]),
        n_layers=r.randint(*E["n_layers"]), n_layers_trans_flow=r.randint(*E["n_layers"]), gin_channels=r.choice(E["gin_channels"]),
        resblock=rb, resblock_kernel_sizes=tuple(r.choice(E["resblock_kernel"]) for _ in range(n_k)),
        resblock_dilation_sizes=tuple(tuple(r.randint(*E["resblock_dilation"]) for _ in range(n_d)) for _ in range(n_k)),
        upsample_rates=tuple(rates), upsample_kernel_sizes=tuple(ks), upsample_initial_channel=init)
    B = r.randint(1, 3)
    lengths = [r.randint(3, 12) for _ in range(B)]
    return hp, lengths, [r.randrange(3) for _ in range(B)], [r.randrange(hp.n_speakers) for _ in range(B)], 100 + i
