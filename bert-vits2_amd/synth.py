"""Seeded synthetic checkpoints and utterances (there are no .pth files and no network here).

Recipe from SURVEY.md §8c/§8d:  weights are drawn per key from a CPU
``torch.Generator`` seeded with ``crc32(key) ^ seed`` so the result does not depend
on iteration order and is identical on every box with this torch build.  Layers
the reference zero-initialises (coupling ``post`` — modules.py:434-435, 558-559;
``ConvFlow.proj`` — modules.py:483-484) are re-randomised so that flows and splines
are exercised, and the Generator is re-scaled so that its output actually depends
on ``z`` (the reference's N(0, 0.01) init — commons.py:6-9 — makes it nearly
constant, a meaningless parity test).
"""
from __future__ import annotations

import math
import zlib
from collections import OrderedDict
from typing import Dict, Optional

import torch

from . import hparams as H
from .schema import param_shapes


def _gen(key: str, seed: int) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def _normal(key, seed, shape, std, mean=0.0):
    return torch.randn(shape, generator=_gen(key, seed), dtype=torch.float32) * std + mean


def synthetic_state_dict(hp: H.HParams, seed: int = 0, pin_durations: Optional[float] = None,
                         dec_post_scale: float = 2.0) -> "OrderedDict[str, torch.Tensor]":
    """A reference-schema fp32 ``state_dict`` with seeded synthetic weights.

    ``pin_durations=2.5`` sets ``dp.proj.weight=0, dp.proj.bias=ln(2.5)`` so that with
    ``sdp_ratio=0`` every symbol gets exactly ceil(2.5)=3 frames (throughput runs).
    """
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for key, shape in param_shapes(hp).items():
        leaf = key.rsplit(".", 1)[-1]
        if leaf in ("gamma",):
            t = _normal(key, seed, shape, 0.1, 1.0)
        elif leaf in ("beta",):
            t = _normal(key, seed, shape, 0.1)
        elif leaf == "bias":
            t = _normal(key, seed, shape, 0.02)
        elif leaf in ("emb_rel_k", "emb_rel_v"):
            t = _normal(key, seed, shape, shape[-1] ** -0.5)
        elif key in ("enc_p.emb.weight", "enc_p.tone_emb.weight", "enc_p.language_emb.weight"):
            t = _normal(key, seed, shape, hp.hidden_channels ** -0.5)
        elif key == "emb_g.weight":
            t = _normal(key, seed, shape, 1.0)
        elif key in ("sdp.flows.0.m", "sdp.flows.0.logs"):
            t = _normal(key, seed, shape, 0.1)
            if key.endswith(".m"):
                t[0, 0] -= 0.9          # centre the stochastic log-duration near ln 2.5
        elif leaf == "weight_g":
            if key.startswith("dec.ups"):
                t = torch.full(shape, 1.0)
            elif key.startswith("dec.resblocks"):
                t = torch.full(shape, 0.4)
            else:                        # WN layers of the residual flow
                t = torch.full(shape, 0.7)
            t = t * (1.0 + 0.05 * torch.randn(shape, generator=_gen(key, seed)))
        elif leaf == "weight_v":
            t = _normal(key, seed, shape, 1.0)
        elif leaf == "weight":
            if len(shape) == 3:
                fan_in = shape[1] * shape[2]
            else:
                fan_in = shape[1]
            std = 1.0 / math.sqrt(fan_in)
            if key.endswith(".post.weight"):          # zero-initialised in the reference
                std = 0.5 / math.sqrt(fan_in)
            elif key.endswith(".proj.weight") and key.startswith("sdp.flows"):
                std = 3.0 / math.sqrt(fan_in)         # zero-initialised in the reference; O(1) spline params
            elif key == "dec.conv_post.weight":
                std = dec_post_scale / math.sqrt(fan_in)
            elif key == "dp.proj.weight":
                std = 0.5 / math.sqrt(fan_in)
            elif ".convs_sep." in key:
                std = 1.0 / math.sqrt(shape[2])
            t = _normal(key, seed, shape, std)
        else:
            raise KeyError(key)
        sd[key] = t.contiguous()
    sd["dp.proj.bias"] = torch.full((1,), 0.9)
    # prior log-std rows of enc_p.proj: trained models sit near logs_p ~ -0.5; keep exp(logs_p) moderate
    C = hp.inter_channels
    sd["enc_p.proj.weight"][C:] *= 0.3
    sd["enc_p.proj.bias"][C:] -= 0.5
    if pin_durations is not None:
        sd["dp.proj.weight"] = torch.zeros_like(sd["dp.proj.weight"])
        sd["dp.proj.bias"] = torch.full((1,), math.log(pin_durations))
    return sd


# ---------------------------------------------------------------------------------------------
# synthetic utterances (SURVEY.md §8d "Synthetic inputs")

_TONE_RANGE = {0: (0, 6), 1: (6, 8), 2: (8, 12)}     # ZH / JP / EN, reference text/symbols.py:73,120,164,178-182


def synthetic_utterance(T: int, index: int = 0, language: int = 0, sid: int = 0, n_speakers: int = 850,
                        base_seed: int = 1234) -> Dict[str, torch.Tensor]:
    """One utterance of T symbols: blanks interspersed at even positions (reference
    commons.py:22-25 / infer.py:113-116), per-language tone ranges, word2ph-style repeated BERT columns
    for the active language and fresh N(0,1) for the two inactive ones (reference infer.py:126-137)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(base_seed + index)
    x = torch.zeros(T, dtype=torch.int64)
    n_odd = T // 2
    x[1::2] = torch.randint(1, 103, (n_odd,), generator=g)
    lo, hi = _TONE_RANGE[language]
    tone = torch.zeros(T, dtype=torch.int64)
    tone[1::2] = torch.randint(lo, hi, (n_odd,), generator=g)
    lang = torch.full((T,), language, dtype=torch.int64)
    berts = []
    for l in range(3):
        if l == language:
            cols = []
            n = 0
            while n < T:
                r = int(torch.randint(2, 4, (1,), generator=g))
                v = torch.randn(H.BERT_DIM, 1, generator=g)
                cols.append(v.expand(H.BERT_DIM, r))
                n += r
            b = torch.cat(cols, 1)[:, :T].contiguous()
        else:
            b = torch.randn(H.BERT_DIM, T, generator=g)
        berts.append(b)
    return dict(x=x, tone=tone, language=lang, bert=berts[0], ja_bert=berts[1], en_bert=berts[2],
                sid=torch.tensor(sid % n_speakers, dtype=torch.int64))


def synthetic_batch(lengths, languages=None, sids=None, first_index: int = 0, n_speakers: int = 850,
                    base_seed: int = 1234) -> Dict[str, torch.Tensor]:
    """Zero-padded batch in the layout of reference data_utils.py collate (BERT as [B,1024,T])."""
    B = len(lengths)
    T = max(lengths)
    languages = languages or [0] * B
    sids = sids or [0] * B
    out = dict(
        x=torch.zeros(B, T, dtype=torch.int64), tone=torch.zeros(B, T, dtype=torch.int64),
        language=torch.zeros(B, T, dtype=torch.int64),
        bert=torch.zeros(B, H.BERT_DIM, T), ja_bert=torch.zeros(B, H.BERT_DIM, T), en_bert=torch.zeros(B, H.BERT_DIM, T),
        x_lengths=torch.tensor(lengths, dtype=torch.int64), sid=torch.zeros(B, dtype=torch.int64),
    )
    for i, (n, l, s) in enumerate(zip(lengths, languages, sids)):
        u = synthetic_utterance(n, first_index + i, l, s, n_speakers, base_seed)
        out["x"][i, :n] = u["x"]
        out["tone"][i, :n] = u["tone"]
        out["language"][i, :n] = u["language"]
        for k in ("bert", "ja_bert", "en_bert"):
            out[k][i, :, :n] = u[k]
        out["sid"][i] = u["sid"]
    return out


def synthetic_noise(B: int, T: int, T_y_cap: int, channels: int = 192, seed: int = 4321):
    """The two RNG draws of reference infer() (models.py:248-251, 1071) as explicit tensors."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    noise_w = torch.randn(B, 2, T, generator=g)
    noise_z = torch.randn(B, channels, T_y_cap, generator=g)
    return noise_w, noise_z
