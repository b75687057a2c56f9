"""BERT feature extraction ON THE DEVICE with libbv2's own kernels (include/bv2_bert.h).

Replaces the model call of the reference's ``text/chinese_bert.py:30-37``::

    models[device] = AutoModelForMaskedLM.from_pretrained("./bert/chinese-roberta-wwm-ext-large").to(device)
    res = models[device](**inputs, output_hidden_states=True)
    res = torch.cat(res["hidden_states"][-3:-2], -1)[0].cpu()

``BertEncoder`` ingests the same checkpoint (``BertForMaskedLM`` / ``BertModel`` ``state_dict``), runs the encoder up to
``hidden_states[-3]`` (the last two layers and the MLM head are never executed) and returns the hidden state as fp32
``[B, hidden, S]`` on the device — the word-level matrix ``bert_features.word_level_feature_cs`` hands to
``SynthesizerTrn.infer(..., bert_index=...)`` without a host round trip, transpose or repeat.  Plumbing only (ctypes + torch device
memory); there is no CPU / PyTorch fallback.  DeBERTa-v2 checkpoints (the reference's Japanese / English extractors) are refused.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import torch

from . import lib as L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class BertEncoder:
    def __init__(self, vocab_size: int = 21128, hidden_size: int = 1024, num_hidden_layers: int = 24, num_attention_heads: int = 16,
                 intermediate_size: int = 4096, max_position_embeddings: int = 512, type_vocab_size: int = 2,
                 layer_norm_eps: float = 1e-12, hidden_state_index: int = -3, hidden_act: str = "gelu",
                 position_embedding_type: str = "absolute", model_type: str = "bert", **_ignored):
        """Arguments are ``BertConfig``'s (pass ``**config.to_dict()``); defaults = chinese-roberta-wwm-ext-large.
        ``hidden_state_index`` is the index into ``hidden_states`` the reference takes (``[-3:-2]``)."""
        if model_type != "bert" or hidden_act != "gelu" or position_embedding_type != "absolute":
            raise NotImplementedError("only BertModel (erf-GELU, absolute positions) is implemented; the reference's Japanese / "
                                      "English extractors are DeBERTa-v2 models")
        n_states = num_hidden_layers + 1
        idx = hidden_state_index if hidden_state_index >= 0 else n_states + hidden_state_index
        if not 1 <= idx <= num_hidden_layers:
            raise ValueError("hidden_state_index must select the output of an encoder layer")
        self.layers_run = idx
        self.hidden_size = hidden_size
        self._lib = L.load()
        cfg = L.BertConfig(C.sizeof(L.BertConfig), vocab_size, hidden_size, num_attention_heads, intermediate_size,
                           max_position_embeddings, type_vocab_size, idx, layer_norm_eps)
        self._cfg = cfg
        self._h = C.c_void_p()
        if self._lib.bv2_bert_create(C.byref(cfg), C.byref(self._h)) != 0:
            raise RuntimeError(self._lib.bv2_bert_last_error(None).decode())
        self._blob: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None         # one workspace, grown to the largest (B, S) seen
        self.device = torch.device("cpu")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.bv2_bert_destroy(self._h)
        except Exception:
            pass

    def _err(self) -> str:
        return self._lib.bv2_bert_last_error(self._h).decode()

    def load_state_dict(self, sd: Mapping[str, torch.Tensor], device="cuda") -> "BertEncoder":
        """Pack a ``BertModel`` / ``BertForMaskedLM`` ``state_dict`` (the ``bert.`` prefix is accepted) and upload it."""
        n = int(self._lib.bv2_bert_packed_bytes(self._h))
        host = torch.zeros(n // 4, dtype=torch.float32)
        for k, v in sd.items():
            if not torch.is_tensor(v) or not v.dtype.is_floating_point:
                continue
            t = v.detach().to("cpu", torch.float32).contiguous()
            shp = (C.c_int64 * max(t.dim(), 1))(*t.shape)
            rc = self._lib.bv2_bert_pack_tensor(self._h, _ptr(host), n, k.encode(), _ptr(t), shp, t.dim())
            if rc < 0:
                raise RuntimeError(self._err())
        if self._lib.bv2_bert_missing(self._h) != 0:
            raise KeyError(self._err())
        dev = torch.device(device)
        if dev.type != "cuda":
            raise RuntimeError("BertEncoder needs a GPU: there is no CPU fallback by design")
        self._blob = host.to(dev)
        self.device = dev
        with torch.cuda.device(dev):
            if self._lib.bv2_bert_attach_weights(self._h, _ptr(self._blob), n) != 0:
                raise RuntimeError(self._err())
        return self

    def replica(self) -> "BertEncoder":
        """A second handle (own workspace, usable on another HIP stream) on the SAME packed weights: request-level concurrency
        (``serving.replicas`` does this for the synthesizer)."""
        if self._blob is None:
            raise RuntimeError("BertEncoder: load_state_dict first")
        r = BertEncoder.__new__(BertEncoder)
        r.layers_run, r.hidden_size, r._lib, r._cfg = self.layers_run, self.hidden_size, self._lib, self._cfg
        r._h = C.c_void_p()
        if self._lib.bv2_bert_create(C.byref(self._cfg), C.byref(r._h)) != 0:
            raise RuntimeError(self._lib.bv2_bert_last_error(None).decode())
        r._blob, r._ws, r.device = self._blob, None, self.device
        with torch.cuda.device(self.device):
            if self._lib.bv2_bert_attach_weights(r._h, _ptr(r._blob), r._blob.numel() * 4) != 0:
                raise RuntimeError(r._err())
        return r

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, token_type_ids: Optional[torch.Tensor] = None,
                 attention_mask: Optional[torch.Tensor] = None, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``hidden_states[hidden_state_index]`` as fp32 ``[B, hidden, S]`` on the device.  ``attention_mask`` must be the prefix
        mask the tokenizer produces (ones, then padding); ``lengths`` [B] may be given instead."""
        if self._blob is None:
            raise RuntimeError("BertEncoder: load_state_dict first")
        dev = self.device
        ids = input_ids.to(dev, torch.int64).contiguous()
        if ids.dim() != 2:
            raise ValueError("input_ids must be [B, S]")
        B, S = ids.shape
        tt = None if token_type_ids is None else token_type_ids.to(dev, torch.int64).contiguous()
        if lengths is None and attention_mask is not None:
            am = attention_mask.to(dev)
            lengths = am.long().sum(1)
            if not bool((am.long() == (torch.arange(S, device=dev)[None, :] < lengths[:, None]).long()).all()):
                raise ValueError("attention_mask must be a prefix mask (right padding)")
        ln = None if lengths is None else lengths.to(dev, torch.int64).contiguous()
        out = torch.empty(B, self.hidden_size, S, dtype=torch.float32, device=dev)
        need = int(self._lib.bv2_bert_workspace_bytes(self._h, B, S))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(need, dtype=torch.uint8, device=dev)
        ws = self._ws
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            rc = self._lib.bv2_bert_forward(self._h, stream, B, S, _ptr(ids), _ptr(tt), _ptr(ln), _ptr(out), _ptr(ws), ws.numel())
        if rc != 0:
            raise RuntimeError(self._err())
        return out
