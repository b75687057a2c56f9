"""Seeded synthetic weights / inputs and the model configurations of the BERT feature extractors (include/bv2_bert.h): there is no
network for the real checkpoints (chinese-roberta-wwm-ext-large, deberta-v2-large-japanese-char-wwm, deberta-v3-large), so bench.py's
BERT legs, smoke() and the tests all draw their ``state_dict`` from here.  Product-side module: a timed leg must not need the checker
(``oracle/``) to exist; the oracles import THESE definitions, so the goldens under tests/golden/ and the device runs see the same
weights bit for bit.  Config values: the ``config.json`` files the reference ships under /root/reference/bert/ (checked by
tests/test_bert_oracle_cpu.py)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

# ---- HuggingFace BertModel (reference text/chinese_bert.py:30-37) ---------------------------------------------------
TINY = dict(vocab_size=97, hidden_size=128, num_hidden_layers=5, num_attention_heads=2, intermediate_size=384,
            max_position_embeddings=48, type_vocab_size=2, layer_norm_eps=1e-12)
MID = dict(vocab_size=211, hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024,
           max_position_embeddings=80, type_vocab_size=2, layer_norm_eps=1e-12)
LARGE = dict(vocab_size=21128, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
             max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)     # chinese-roberta-wwm-ext-large's config.json



def bert_state_dict(cfg: Dict, seed: int = 0, layers: Optional[int] = None) -> Dict[str, torch.Tensor]:
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



# ---- DebertaV2Model / DebertaV2ForMaskedLM (reference text/japanese_bert.py:34-43, text/english_bert_mock.py:30-41) -----
TINY_V3 = dict(vocab_size=131, hidden_size=128, num_hidden_layers=5, num_attention_heads=2, intermediate_size=384,
               max_position_embeddings=64, relative_attention=True, position_buckets=16, norm_rel_ebd="layer_norm", share_att_key=True,
               pos_att_type="p2c|c2p", layer_norm_eps=1e-7, max_relative_positions=-1, position_biased_input=False, type_vocab_size=0)
TINY_JA = dict(TINY_V3, vocab_size=97, conv_kernel_size=3, conv_act="gelu", pos_att_type=["p2c", "c2p"], attention_head_size=64)
MID_V3 = dict(TINY_V3, vocab_size=211, hidden_size=256, num_hidden_layers=4, num_attention_heads=4, intermediate_size=1024,
              max_position_embeddings=160, position_buckets=32)
# /root/reference/bert/deberta-v3-large/config.json and deberta-v2-large-japanese-char-wwm/config.json
LARGE_V3 = dict(vocab_size=128100, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
                max_position_embeddings=512, relative_attention=True, position_buckets=256, norm_rel_ebd="layer_norm", share_att_key=True,
                pos_att_type="p2c|c2p", layer_norm_eps=1e-7, max_relative_positions=-1, position_biased_input=False, type_vocab_size=0)
LARGE_JA = dict(LARGE_V3, vocab_size=22012, conv_kernel_size=3, conv_act="gelu", pos_att_type=["p2c", "c2p"], attention_head_size=64)


def att_span(cfg: Dict) -> int:
    mr = cfg.get("max_relative_positions", -1)
    mr = cfg["max_position_embeddings"] if mr < 1 else mr
    pb = cfg.get("position_buckets", -1)
    return pb if pb > 0 else mr



def deberta_state_dict(cfg: Dict, seed: int = 0, layers: Optional[int] = None) -> Dict[str, torch.Tensor]:
    """Seeded synthetic ``DebertaV2Model.state_dict()``; same scale choices as the BERT one (bert_state_dict above)."""
    g = torch.Generator().manual_seed(2000 + seed)
    C, I = cfg["hidden_size"], cfg["intermediate_size"]
    n = cfg["num_hidden_layers"] if layers is None else layers
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g) * sc
    sd = {"embeddings.word_embeddings.weight": r(cfg["vocab_size"], C, sc=0.8),
          "embeddings.LayerNorm.weight": 1 + r(C, sc=0.1), "embeddings.LayerNorm.bias": r(C, sc=0.1),
          "encoder.rel_embeddings.weight": r(2 * att_span(cfg), C, sc=0.8),
          "encoder.LayerNorm.weight": 1 + r(C, sc=0.1), "encoder.LayerNorm.bias": r(C, sc=0.1)}
    k = cfg.get("conv_kernel_size", 0)
    if k > 0:
        sd["encoder.conv.conv.weight"] = r(C, C, k, sc=1.0 / math.sqrt(C * k))
        sd["encoder.conv.conv.bias"] = r(C, sc=0.05)
        sd["encoder.conv.LayerNorm.weight"] = 1 + r(C, sc=0.1)
        sd["encoder.conv.LayerNorm.bias"] = r(C, sc=0.1)
    for i in range(n):
        p = f"encoder.layer.{i}."
        for name, (o, c_in, sc) in {"attention.self.query_proj": (C, C, 2.0), "attention.self.key_proj": (C, C, 2.0),
                                    "attention.self.value_proj": (C, C, 1.0), "attention.output.dense": (C, C, 1.0),
                                    "intermediate.dense": (I, C, 1.0), "output.dense": (C, I, 1.0)}.items():
            sd[p + name + ".weight"] = r(o, c_in, sc=sc / math.sqrt(c_in))
            sd[p + name + ".bias"] = r(o, sc=0.05)
        for name in ("attention.output.LayerNorm", "output.LayerNorm"):
            sd[p + name + ".weight"] = 1 + r(C, sc=0.1)
            sd[p + name + ".bias"] = r(C, sc=0.1)
    return sd
