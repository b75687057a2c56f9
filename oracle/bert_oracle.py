"""TEST INFRASTRUCTURE — CPU restatement of HuggingFace ``BertModel``'s forward pass up to ``hidden_states[n]``, the oracle for the
device BERT feature extractor (include/bv2_bert.h, bert-vits2_amd/csrc/bv2_bert.cpp).  Only tests/, bench.py's cpu_baseline
leg and __graft_entry__.smoke() may import this; the product path never does.

The algorithm lives in a third-party dependency of the reference that is not under /root/reference: ``transformers`` (listed
unpinned in the reference's requirements.txt:11; 5.15.0 in the build image).  Reference call site: text/chinese_bert.py:30-37
(``AutoModelForMaskedLM.from_pretrained(...)``, ``output_hidden_states=True``, ``res["hidden_states"][-3:-2]``).  What is restated,
by class of transformers' ``models/bert/modeling_bert.py``:

* ``BertEmbeddings.forward``      word + token_type + position embeddings (absolute, ids 0..S-1), LayerNorm(eps), dropout = id
* ``BertSelfAttention.forward``   q/k/v Linear, [B, heads, S, d], scores / sqrt(d) + additive mask (0 / finfo.min), softmax, P V
* ``BertSelfOutput.forward``      dense, dropout = id, LayerNorm(dense + input)
* ``BertIntermediate.forward``    dense, erf GELU
* ``BertOutput.forward``          dense, dropout = id, LayerNorm(dense + attention_output)
* ``BertEncoder.forward``         hidden_states = (embedding output, layer 1 output, ..., layer N output); [-3] = layer N-2's output

PINNED by tests/golden/bert_*.npz, produced by the REAL ``transformers.BertModel`` on seeded synthetic weights
(oracle/gen_bert_golden.py; tests/test_bert_oracle_cpu.py holds this restatement to them).
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

TINY = dict(vocab_size=97, hidden_size=128, num_hidden_layers=5, num_attention_heads=2, intermediate_size=384,
            max_position_embeddings=48, type_vocab_size=2, layer_norm_eps=1e-12)
MID = dict(vocab_size=211, hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024,
           max_position_embeddings=80, type_vocab_size=2, layer_norm_eps=1e-12)
LARGE = dict(vocab_size=21128, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
             max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)     # chinese-roberta-wwm-ext-large's config.json


def synthetic_state_dict(cfg: Dict, seed: int = 0, layers: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Seeded synthetic ``BertModel.state_dict()`` (no pooler): there is no network for the real checkpoint.  Scales are chosen so
    that attention is far from uniform and LayerNorm inputs have O(1) spread (the pretrained model's regime)."""
    g = torch.Generator().manual_seed(1000 + seed)
    C, I = cfg["hidden_size"], cfg["intermediate_size"]
    n = cfg["num_hidden_layers"] if layers is None else layers
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sd = {"embeddings.word_embeddings.weight": r(cfg["vocab_size"], C, sc=0.6),
          "embeddings.position_embeddings.weight": r(cfg["max_position_embeddings"], C, sc=0.3),
          "embeddings.token_type_embeddings.weight": r(cfg["type_vocab_size"], C, sc=0.2),
          "embeddings.LayerNorm.weight": 1 + r(C, sc=0.1), "embeddings.LayerNorm.bias": r(C, sc=0.1)}
    for i in range(n):
        p = f"encoder.layer.{i}."
        for name, (o, c_in, sc) in {"attention.self.query": (C, C, 2.0), "attention.self.key": (C, C, 2.0),
                                    "attention.self.value": (C, C, 1.0), "attention.output.dense": (C, C, 1.0),
                                    "intermediate.dense": (I, C, 1.0), "output.dense": (C, I, 1.0)}.items():
            sd[p + name + ".weight"] = r(o, c_in, sc=sc / math.sqrt(c_in))
            sd[p + name + ".bias"] = r(o, sc=0.05)
        for name in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + name + ".weight"] = 1 + r(C, sc=0.1)
            sd[p + name + ".bias"] = r(C, sc=0.1)
    return sd


def synthetic_inputs(cfg: Dict, lengths, seed: int = 0):
    g = torch.Generator().manual_seed(77 + seed)
    S = max(lengths)
    ids = torch.randint(0, cfg["vocab_size"], (len(lengths), S), generator=g)
    for b, n in enumerate(lengths):
        ids[b, n:] = 0                                    # [PAD]
    return ids, torch.tensor(lengths, dtype=torch.int64)


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def hidden_state(sd: Dict[str, torch.Tensor], cfg: Dict, input_ids: torch.Tensor, n_layers: int,
                 token_type_ids: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None,
                 dtype=torch.float32) -> torch.Tensor:
    """``BertModel(input_ids, token_type_ids, attention_mask).hidden_states[n_layers]`` as [B, S, C]."""
    sd = {k[5:] if k.startswith("bert.") else k: v.to(dtype) for k, v in sd.items() if torch.is_tensor(v) and v.dtype.is_floating_point}
    B, S = input_ids.shape
    H, C = cfg["num_attention_heads"], cfg["hidden_size"]
    d = C // H
    eps = cfg["layer_norm_eps"]
    tt = torch.zeros_like(input_ids) if token_type_ids is None else token_type_ids
    x = sd["embeddings.word_embeddings.weight"][input_ids] + sd["embeddings.token_type_embeddings.weight"][tt]
    x = x + sd["embeddings.position_embeddings.weight"][torch.arange(S)][None]
    x = _ln(x, sd["embeddings.LayerNorm.weight"], sd["embeddings.LayerNorm.bias"], eps)
    add_mask = None
    if lengths is not None:
        valid = torch.arange(S)[None, :] < lengths[:, None]
        add_mask = torch.zeros(B, 1, 1, S, dtype=dtype).masked_fill(~valid[:, None, None, :], torch.finfo(dtype).min)
    for i in range(n_layers):
        p = f"encoder.layer.{i}."
        lin = lambda name, t: F.linear(t, sd[p + name + ".weight"], sd[p + name + ".bias"])
        split = lambda t: t.view(B, S, H, d).transpose(1, 2)
        q, k, v = split(lin("attention.self.query", x)), split(lin("attention.self.key", x)), split(lin("attention.self.value", x))
        sc = q @ k.transpose(-1, -2) / math.sqrt(d)
        if add_mask is not None:
            sc = sc + add_mask
        ctx = (torch.softmax(sc, -1) @ v).transpose(1, 2).reshape(B, S, C)
        x1 = _ln(lin("attention.output.dense", ctx) + x, sd[p + "attention.output.LayerNorm.weight"],
                 sd[p + "attention.output.LayerNorm.bias"], eps)
        h = F.gelu(lin("intermediate.dense", x1))
        x = _ln(lin("output.dense", h) + x1, sd[p + "output.LayerNorm.weight"], sd[p + "output.LayerNorm.bias"], eps)
    return x
