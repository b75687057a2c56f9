"""BERT feature extraction kept on the GPU — the step right in front of ``SynthesizerTrn.infer``.

The reference (text/chinese_bert.py:15-60, japanese_bert.py:16-65, english_bert_mock.py:15-60) runs the HF model, takes
``hidden_states[-3]``, copies it to the HOST (``.cpu()`` at chinese_bert.py:37), repeats word ``i``'s row ``word2ph[i]`` times in a
Python loop (:48-58), transposes, and ``infer.get_text`` uploads the ``[1024, T]`` result again (infer.py:107-152).  Here

* the hidden state never leaves the device;
* the repeat is not materialised: ``word_level_feature`` returns the word-level matrix ``[1024, S]`` plus an int32 index ``[T]``
  (symbol t -> word), and the TextEncoder front of ``libbv2.so`` gathers through it (``bv2_encode_in.bert_index``): per
  utterance 1024*S*4 bytes cross HBM instead of 1024*T*4 (T ~ 2.5 S with blanks interspersed), and no repeat kernel runs;
* ``style_text`` mixing (chinese_bert.py:38-47, 52-56) is applied at word level on the device — mixing and repeating commute.

The BERT encoder itself: ``bert_encoder.BertEncoder`` runs the Chinese extractor (a HuggingFace ``BertModel``) with libbv2's own
kernels and hands back ``[1024, S]`` directly (``get_bert_feature`` below takes it as ``model``) — with ``model_type="deberta-v2"``
the same class runs the reference's Japanese / English extractors (DeBERTa-v2 / v3 large); any HF model object works as well
(PyTorch-ROCm library kernels).  ``len(word2ph) == S`` is checked on the host, so every index is < S by construction; the device
gather additionally clamps to the feature's own column count.
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple

import torch


def word_to_symbol_index(word2ph: Sequence[int], device=None) -> torch.Tensor:
    """int32 [T]: the word each symbol belongs to — the index form of the reference's repeat loop (chinese_bert.py:48-58).
    Built with numpy (a few microseconds; the torch CPU ops it replaces cost milliseconds per call on a many-core host) and
    uploaded in one small copy."""
    import numpy as np
    w = np.asarray(list(word2ph), dtype=np.int64)
    if (w < 0).any():
        raise ValueError("word2ph entries must be >= 0")
    idx = torch.from_numpy(np.repeat(np.arange(len(w), dtype=np.int32), w))
    return idx if device is None else idx.to(device, non_blocking=True)


def word_level_feature(hidden: torch.Tensor, word2ph: Sequence[int], style_hidden: Optional[torch.Tensor] = None,
                       style_weight: float = 0.7) -> Tuple[torch.Tensor, torch.Tensor]:
    """``hidden`` [S, 1024] = hidden_states[-3][0] of the BERT model (any device) -> (feature [1024, S], index [T] int32), both on
    ``hidden``'s device.  ``len(word2ph)`` must equal S (the reference asserts ``len(word2ph) == len(text) + 2``, :42)."""
    if hidden.dim() != 2 or hidden.shape[0] != len(word2ph):
        raise ValueError(f"hidden must be [len(word2ph)={len(word2ph)}, C], got {tuple(hidden.shape)}")
    res = hidden.float()
    if style_hidden is not None:
        res = res * (1 - style_weight) + style_hidden.float().mean(0, keepdim=True) * style_weight
    return res.t().contiguous(), word_to_symbol_index(word2ph, hidden.device)


def word_level_feature_cs(feature_cs: torch.Tensor, word2ph: Sequence[int], style_cs: Optional[torch.Tensor] = None,
                          style_weight: float = 0.7) -> Tuple[torch.Tensor, torch.Tensor]:
    """Same as ``word_level_feature`` for a hidden state that already is channels-first ``[1024, S]`` (``BertEncoder``'s output):
    no transpose at all."""
    if feature_cs.dim() != 2 or feature_cs.shape[1] != len(word2ph):
        raise ValueError(f"feature must be [C, len(word2ph)={len(word2ph)}], got {tuple(feature_cs.shape)}")
    res = feature_cs.float()
    if style_cs is not None:
        res = res * (1 - style_weight) + style_cs.float().mean(1, keepdim=True) * style_weight
    return res.contiguous(), word_to_symbol_index(word2ph, feature_cs.device)


def expand(feature: torch.Tensor, index: torch.Tensor) -> torch.Tensor:
    """What the reference's ``get_bert_feature`` returns: the symbol-level ``[1024, T]`` matrix (tests / callers that want it)."""
    return feature.index_select(1, index.long())


@torch.no_grad()
def get_bert_feature(text: str, word2ph: Sequence[int], tokenizer, model, device, style_text: Optional[str] = None,
                     style_weight: float = 0.7) -> Tuple[torch.Tensor, torch.Tensor]:
    """reference text/chinese_bert.py:15-60 (same arguments plus the tokenizer / model the reference keeps in module globals), on
    ``device``, returning the word-level pair instead of the repeated matrix."""
    from .bert_encoder import BertEncoder
    if isinstance(model, BertEncoder):                # libbv2's own BERT: [1024, S] on the device, nothing to transpose
        run_cs = lambda t: model(**{k: v.to(device) for k, v in tokenizer(t, return_tensors="pt").items()})[0]
        return word_level_feature_cs(run_cs(text), word2ph, run_cs(style_text) if style_text else None, style_weight)
    run = lambda t: model(**{k: v.to(device) for k, v in tokenizer(t, return_tensors="pt").items()},
                          output_hidden_states=True)["hidden_states"][-3][0]
    res = run(text)
    style = run(style_text) if style_text else None
    return word_level_feature(res, word2ph, style, style_weight)


def batch_index(indices: Sequence[torch.Tensor], T: int, device) -> torch.Tensor:
    """Pad per-utterance index vectors to ``[B, T]`` (padded symbols point at word 0; they are masked out by x_lengths)."""
    out = torch.zeros(len(indices), T, dtype=torch.int32, device=device)
    for i, ix in enumerate(indices):
        out[i, : ix.numel()] = ix.to(device)
    return out
