"""TEST INFRASTRUCTURE — CPU restatement of HuggingFace ``DebertaV2Model``'s forward pass up to ``hidden_states[n]``: the oracle for
the DeBERTa-v2 form of the device BERT feature extractor (include/bv2_bert.h, arch = BV2_BERT_ARCH_DEBERTA_V2).  Only tests/,
bench.py's cpu_baseline leg and __graft_entry__.smoke() may import this; the product path never does.

Reference call sites: text/japanese_bert.py:34-43 (``AutoModelForMaskedLM.from_pretrained("./bert/deberta-v2-large-japanese-char-wwm")``,
``output_hidden_states=True``, ``hidden_states[-3:-2]``) and text/english_bert_mock.py:30-41 (``DebertaV2Model.from_pretrained(
"./bert/deberta-v3-large")``); their configs ship with the reference (/root/reference/bert/deberta-v2-large-japanese-char-wwm/config.json,
/root/reference/bert/deberta-v3-large/config.json: 24 x 1024, 16 heads, relative_attention, position_buckets 256, share_att_key,
pos_att_type p2c|c2p, norm_rel_ebd layer_norm, position_biased_input false, type_vocab_size 0, layer_norm_eps 1e-7; the Japanese
model adds conv_kernel_size 3 / conv_act gelu).  The algorithm lives in ``transformers`` (not under /root/reference; unpinned in
requirements.txt:11, 5.15.0 in the build image); restated here by class of ``models/deberta_v2/modeling_deberta_v2.py``:

* ``DebertaV2Embeddings.forward``        word embeddings only (no position / token-type input), LayerNorm, * mask
* ``make_log_bucket_position`` / ``build_relative_position``   relative position q - k, exact up to bucket/2, log-spaced beyond
* ``DebertaV2Encoder.get_rel_embedding`` LayerNorm(rel_embeddings.weight);  ``get_attention_mask``: outer product of the token mask
* ``DisentangledSelfAttention``          scores = (Q K^T + c2p + p2c) / sqrt(3 d): c2p[i][j] = Q_i . PK[clamp(rel(i,j) + span)],
                                         p2c[i][j] = K_j . PQ[clamp(-rel(j,i) + span)], PK / PQ = key_proj / query_proj(rel_embeddings)
                                         (share_att_key); masked_fill(finfo.min), softmax, P V
* ``DebertaV2SelfOutput`` / ``DebertaV2Intermediate`` / ``DebertaV2Output``   as in BERT (erf GELU, post-LayerNorm)
* ``ConvLayer`` (after layer 0 only)     LayerNorm(layer0_out + act(conv1d_k(embeddings) masked)) * mask

PINNED by tests/golden/deberta_*.npz, produced by the REAL transformers classes the reference instantiates on seeded synthetic weights
(oracle/gen_bert_golden.py; tests/test_bert_oracle_cpu.py holds this restatement to them).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from bert_vits2_amd.bert_synth import (LARGE_JA, LARGE_V3, MID_V3, TINY_JA, TINY_V3, att_span,  # noqa: F401
                                       deberta_state_dict as synthetic_state_dict)

def log_bucket_position(rel: torch.Tensor, bucket_size: int, max_position: int) -> torch.Tensor:
    """``make_log_bucket_position``: float32 arithmetic exactly as transformers does it (the ceil() makes it rounding-sensitive)."""
    sign = torch.sign(rel)
    mid = bucket_size // 2
    abs_pos = torch.where((rel < mid) & (rel > -mid), torch.tensor(mid - 1).type_as(rel), torch.abs(rel))
    log_pos = torch.ceil(torch.log(abs_pos / mid) / torch.log(torch.tensor((max_position - 1) / mid)) * (mid - 1)) + mid
    return torch.where(abs_pos <= mid, rel.type_as(log_pos), log_pos * sign)


def relative_index_table(cfg: Dict, max_len: int) -> torch.Tensor:
    """int64 [2*max_len - 1]: entry (q - k) + max_len - 1 = clamp(bucket(q - k) + span, 0, 2 span - 1) — the one index both the
    content->position and the position->content term gather with (HF's p2c index clamp(-bucket(k - q) + span) is the same
    number: the bucket function is odd).  This table is also what the device kernel reads (packed by the wrapper)."""
    span = att_span(cfg)
    mr = cfg.get("max_relative_positions", -1)
    mr = cfg["max_position_embeddings"] if mr < 1 else mr
    rel = torch.arange(-(max_len - 1), max_len, dtype=torch.long)
    pb = cfg.get("position_buckets", -1)
    b = log_bucket_position(rel, pb, mr).to(torch.long) if pb > 0 else rel
    return torch.clamp(b + span, 0, 2 * span - 1)


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def hidden_state(sd: Dict[str, torch.Tensor], cfg: Dict, input_ids: torch.Tensor, n_layers: int,
                 lengths: Optional[torch.Tensor] = None, dtype=torch.float32) -> torch.Tensor:
    """``DebertaV2Model(input_ids, attention_mask).hidden_states[n_layers]`` as [B, S, C]."""
    sd = {(k[8:] if k.startswith("deberta.") else k): v.to(dtype) for k, v in sd.items() if torch.is_tensor(v) and v.dtype.is_floating_point}
    B, S = input_ids.shape
    H, C = cfg["num_attention_heads"], cfg["hidden_size"]
    d = C // H
    eps = cfg["layer_norm_eps"]
    span = att_span(cfg)
    mask = torch.ones(B, S, dtype=dtype) if lengths is None else (torch.arange(S)[None, :] < lengths[:, None]).to(dtype)
    x = _ln(sd["embeddings.word_embeddings.weight"][input_ids], sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    x = x * mask[..., None]
    emb = x
    rel = _ln(sd["encoder.rel_embeddings.weight"], sd["encoder.LayerNorm.weight"], sd["encoder.LayerNorm.bias"], eps)[: 2 * span]
    # relative positions exactly as build_relative_position + the two clamps of disentangled_attention_bias
    mr = cfg.get("max_relative_positions", -1)
    mr = cfg["max_position_embeddings"] if mr < 1 else mr
    ids = torch.arange(S, dtype=torch.long)
    rel_pos = ids[:, None] - ids[None, :]
    pb = cfg.get("position_buckets", -1)
    if pb > 0:
        rel_pos = log_bucket_position(rel_pos, pb, mr).to(torch.long)
    c2p_pos = torch.clamp(rel_pos + span, 0, 2 * span - 1)                    # [q, k]
    p2c_pos = torch.clamp(-rel_pos + span, 0, 2 * span - 1)                   # [k(as query axis of r_pos), k]
    amask = (mask[:, None, :, None] * mask[:, None, None, :]).bool()
    scale = math.sqrt(d * 3)
    for i in range(n_layers):
        p = f"encoder.layer.{i}."
        lin = lambda name, t: F.linear(t, sd[p + name + ".weight"], sd[p + name + ".bias"])
        split = lambda t: t.view(t.shape[0], t.shape[1], H, d).transpose(1, 2)
        q, k, v = split(lin("attention.self.query_proj", x)), split(lin("attention.self.key_proj", x)), split(lin("attention.self.value_proj", x))
        pq = split(lin("attention.self.query_proj", rel[None]))                # [1, H, 2 span, d]
        pk = split(lin("attention.self.key_proj", rel[None]))
        sc = q @ (k.transpose(-1, -2) / scale)
        c2p = torch.gather(q @ pk.transpose(-1, -2), -1, c2p_pos[None, None].expand(B, H, S, S)) / scale
        p2c = torch.gather(k @ pq.transpose(-1, -2), -1, p2c_pos[None, None].expand(B, H, S, S)).transpose(-1, -2) / scale
        sc = (sc + (c2p + p2c)).masked_fill(~amask, torch.finfo(dtype).min)
        ctx = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, S, C)
        a = _ln(lin("attention.output.dense", ctx) + x, sd[p + "attention.output.LayerNorm.weight"], sd[p + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(lin("intermediate.dense", a))
        y = _ln(lin("output.dense", h) + a, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
        if i == 0 and cfg.get("conv_kernel_size", 0) > 0:
            kk = cfg["conv_kernel_size"]
            out = F.conv1d(emb.transpose(1, 2), sd["encoder.conv.conv.weight"], sd["encoder.conv.conv.bias"], padding=(kk - 1) // 2).transpose(1, 2)
            out = out * mask[..., None]
            act = F.gelu if cfg.get("conv_act", "tanh") == "gelu" else torch.tanh
            y = _ln(y + act(out), sd["encoder.conv.LayerNorm.weight"], sd["encoder.conv.LayerNorm.bias"], eps) * mask[..., None]
        x = y
    return x
