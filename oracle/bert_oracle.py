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

from bert_vits2_amd.bert_synth import LARGE, MID, TINY, bert_state_dict as synthetic_state_dict, synthetic_inputs  # noqa: F401  (shared with the product-side legs)

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
