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
