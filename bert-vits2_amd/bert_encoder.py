"""BERT feature extraction ON THE DEVICE with libbv2's own kernels (include/bv2_bert.h).

Replaces the model call of the reference's ``text/chinese_bert.py:30-37``::

    models[device] = AutoModelForMaskedLM.from_pretrained("./bert/chinese-roberta-wwm-ext-large").to(device)
    res = models[device](**inputs, output_hidden_states=True)
    res = torch.cat(res["hidden_states"][-3:-2], -1)[0].cpu()

``BertEncoder`` ingests the same checkpoint (``BertForMaskedLM`` / ``BertModel`` ``state_dict``), runs the encoder up to
``hidden_states[-3]`` (the last two layers and the MLM head are never executed) and returns the hidden state as fp32
``[B, hidden, S]`` on the device — the word-level matrix ``bert_features.word_level_feature_cs`` hands to
``SynthesizerTrn.infer(..., bert_index=...)`` without a host round trip, transpose or repeat.  Plumbing only (ctypes + torch device
memory); there is no CPU / PyTorch fallback.  With ``model_type="deberta-v2"`` the same class runs the reference's Japanese / English
extractors (``text/japanese_bert.py:34-43`` ``DebertaV2ForMaskedLM``, ``text/english_bert_mock.py:30-41`` ``DebertaV2Model``):
disentangled attention with log-bucket relative positions and, for the Japanese model, the ConvLayer behind the first layer.
"""
from __future__ import annotations

import ctypes as C
from typing import Mapping, Optional

import torch

from . import lib as L


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _relative_index_table(position_buckets: int, max_relative_positions: int, span: int, max_len: int) -> torch.Tensor:
    """fp32 [2*max_len - 1]: clamp(bucket(r) + span, 0, 2*span - 1) for r = -(max_len-1) .. max_len-1 — transformers'
    ``make_log_bucket_position`` (models/deberta_v2/modeling_deberta_v2.py) evaluated with the same float32 torch ops (its ceil()
    makes the table rounding-sensitive), followed by the clamp of ``disentangled_attention_bias``.  Both relative terms gather with
    this one index: the bucket function is odd."""
    rel = torch.arange(-(max_len - 1), max_len, dtype=torch.long)
    if position_buckets > 0:
        sign = torch.sign(rel)
        mid = position_buckets // 2
        abs_pos = torch.where((rel < mid) & (rel > -mid), torch.tensor(mid - 1).type_as(rel), torch.abs(rel))
        log_pos = torch.ceil(torch.log(abs_pos / mid) / torch.log(torch.tensor((max_relative_positions - 1) / mid)) * (mid - 1)) + mid
        rel = torch.where(abs_pos <= mid, rel.type_as(log_pos), log_pos * sign).to(torch.long)
    return torch.clamp(rel + span, 0, 2 * span - 1).to(torch.float32)


class BertEncoder:
    def __init__(self, vocab_size: int = 21128, hidden_size: int = 1024, num_hidden_layers: int = 24, num_attention_heads: int = 16,
                 intermediate_size: int = 4096, max_position_embeddings: int = 512, type_vocab_size: int = 2,
                 layer_norm_eps: float = 1e-12, hidden_state_index: int = -3, hidden_act: str = "gelu",
                 position_embedding_type: str = "absolute", model_type: str = "bert", **extra):
        """Arguments are the HF config's (pass ``**config.to_dict()``); defaults = chinese-roberta-wwm-ext-large
        (/root/reference/bert/chinese-roberta-wwm-ext-large/config.json).  ``model_type="deberta-v2"`` selects the DeBERTa-v2 form
        (the reference's Japanese / English extractors) and reads ``position_buckets``, ``max_relative_positions``,
        ``conv_kernel_size`` / ``conv_act`` from the same dict.  ``hidden_state_index`` is the index into ``hidden_states`` the
        reference takes (``[-3:-2]``)."""
        if hidden_act != "gelu":
            raise NotImplementedError("only erf-GELU models are implemented")
        self.arch = {"bert": 0, "deberta-v2": 1}.get(model_type)
        if self.arch is None:
            raise NotImplementedError(f"model_type {model_type!r}: only 'bert' (BertModel) and 'deberta-v2' (DebertaV2Model) are implemented")
        span, conv_k = 0, 0
        if self.arch == 0:
            if position_embedding_type != "absolute":
                raise NotImplementedError("BertModel with relative position embeddings is not implemented")
        else:
            pat = extra.get("pos_att_type") or []
            pat = sorted(pat.split("|")) if isinstance(pat, str) else sorted(pat)
            if not (extra.get("relative_attention") and extra.get("share_att_key") and pat == ["c2p", "p2c"]
                    and "layer_norm" in str(extra.get("norm_rel_ebd", "none")).lower() and not extra.get("position_biased_input", True)
                    and type_vocab_size == 0 and extra.get("embedding_size", hidden_size) == hidden_size
                    and extra.get("attention_head_size", hidden_size // num_attention_heads) == hidden_size // num_attention_heads):
                raise NotImplementedError("DeBERTa-v2 variant not implemented: need relative_attention, share_att_key, pos_att_type "
                                          "c2p|p2c, norm_rel_ebd layer_norm, position_biased_input false, type_vocab_size 0 (the "
                                          "reference's deberta-v3-large / deberta-v2-large-japanese-char-wwm configs)")
            self._max_rel = extra.get("max_relative_positions", -1)
            if self._max_rel < 1:
                self._max_rel = max_position_embeddings
            self._buckets = extra.get("position_buckets", -1)
            span = self._buckets if self._buckets > 0 else self._max_rel
            conv_k = int(extra.get("conv_kernel_size", 0) or 0)
            if conv_k > 0 and extra.get("conv_act", "tanh") != "gelu":
                raise NotImplementedError("ConvLayer with conv_act other than gelu is not implemented")
            if conv_k > 0 and extra.get("conv_groups", 1) != 1:
                raise NotImplementedError("grouped ConvLayer is not implemented")
        n_states = num_hidden_layers + 1
        idx = hidden_state_index if hidden_state_index >= 0 else n_states + hidden_state_index
        if not 1 <= idx <= num_hidden_layers:
            raise ValueError("hidden_state_index must select the output of an encoder layer")
        self.layers_run = idx
        self.hidden_size = hidden_size
        self._span, self._max_pos, self._eps = span, max_position_embeddings, layer_norm_eps
        self._lib = L.load()
        cfg = L.BertConfig(C.sizeof(L.BertConfig), vocab_size, hidden_size, num_attention_heads, intermediate_size,
                           max_position_embeddings, type_vocab_size, idx, layer_norm_eps, self.arch, span, conv_k)
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
        if self.arch == 1:
            sd = self._with_deberta_derived(sd)
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

    def _with_deberta_derived(self, sd: Mapping[str, torch.Tensor]) -> dict:
        """The three weight-only tensors of include/bv2_bert.h: per layer pos_key / pos_query = key_proj / query_proj applied to
        LayerNorm(encoder.rel_embeddings.weight) (DebertaV2Encoder.get_rel_embedding + the share_att_key branch of
        disentangled_attention_bias — they do not depend on the input, so they are computed once here instead of per forward), and
        the relative index table."""
        import torch.nn.functional as F
        sd = {(k[8:] if k.startswith("deberta.") else k): v for k, v in sd.items()}
        f = lambda k: sd[k].detach().to("cpu", torch.float32)
        rel = F.layer_norm(f("encoder.rel_embeddings.weight"), (self.hidden_size,), f("encoder.LayerNorm.weight"),
                           f("encoder.LayerNorm.bias"), self._eps)[: 2 * self._span]
        out = dict(sd)
        for i in range(self.layers_run):
            p = f"encoder.layer.{i}.attention.self."
            out[p + "pos_key"] = F.linear(rel, f(p + "key_proj.weight"), f(p + "key_proj.bias"))
            out[p + "pos_query"] = F.linear(rel, f(p + "query_proj.weight"), f(p + "query_proj.bias"))
        out["encoder.relative_index"] = _relative_index_table(self._buckets, self._max_rel, self._span, self._max_pos)
        return out

    def set_option(self, key: str, value: int) -> None:
        """``bv2_bert_set_option`` (include/bv2_bert.h): kernel-selection switches for A/B measurements, e.g. ``"prefetch"``."""
        if self._lib.bv2_bert_set_option(self._h, key.encode(), int(value)) != 0:
            raise ValueError(self._err())

    def replica(self) -> "BertEncoder":
        """A second handle (own workspace, usable on another HIP stream) on the SAME packed weights: request-level concurrency
        (``serving.replicas`` does this for the synthesizer)."""
        if self._blob is None:
            raise RuntimeError("BertEncoder: load_state_dict first")
        r = BertEncoder.__new__(BertEncoder)
        r.layers_run, r.hidden_size, r._lib, r._cfg, r.arch = self.layers_run, self.hidden_size, self._lib, self._cfg, self.arch
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
            # the tokenizer's mask is a HOST tensor: lengths and the prefix check are computed there, before the upload — a device
            # tensor is taken on trust (checking it would be a device-to-host sync per sentence in front of infer())
            am = attention_mask
            lengths = am.long().sum(1)
            if am.device.type == "cpu":
                if not bool((am.long() == (torch.arange(S)[None, :] < lengths[:, None]).long()).all()):
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
