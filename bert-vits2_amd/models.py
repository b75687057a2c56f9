"""Drop-in ``SynthesizerTrn`` whose ``infer()`` runs on the hand-written gfx950 kernels of ``libbv2.so``.

Mirrors the surface of reference ``models.SynthesizerTrn`` that its callers use
(reference infer.py:84-104, 301-314; train_ms.py:772-784; utils.py:65-120):

* the constructor signature (reference models.py:816-842), unknown ``**kwargs`` swallowed;
* ``.to(device)`` / ``.eval()`` / ``.state_dict()`` / ``.load_state_dict(strict=False)`` with the reference's key
  schema, so reference ``utils.load_checkpoint`` works unchanged;
* ``.infer(x, x_lengths, sid, tone, language, bert, ja_bert, en_bert, noise_scale=.667, length_scale=1,
  noise_scale_w=0.8, max_len=None, sdp_ratio=0, y=None)`` → ``(o, attn, y_mask, (z, z_p, m_p, logs_p))``
  (reference models.py:1026-1074), same shapes/dtypes.

This module is host plumbing only: PyTorch supplies device memory, the current HIP stream and the RNG; every FLOP of
``infer()`` happens inside the C-ABI library.  There is deliberately NO PyTorch/CPU fallback: without a GPU or without
the built library ``infer()`` raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
from torch import nn

from . import hparams as H
from . import lib as L
from .schema import param_shapes


def draw_noise_w(B: int, T: int, device) -> torch.Tensor:
    """Draw #1 of reference infer(), models.py:248-251: ``torch.randn(B, 2, T).to(device=..., dtype=...)`` — taken from the
    global CPU generator even when the model sits on a GPU, then uploaded.  Same call, same stream position: a seeded
    reference run and a seeded run of this shim draw the same SDP noise (RNG contract)."""
    return torch.randn(B, 2, T).to(device=device, dtype=torch.float32)


def draw_noise_z(B: int, C: int, Ty: int, device) -> torch.Tensor:
    """Draw #2, models.py:1071: ``torch.randn_like(m_p)`` where ``m_p`` is a TRANSPOSED view ([B,C,T_y] with strides
    (C*T_y, 1, C), models.py:1064-1066) on the model's device.  ``randn_like`` fills in memory order, so only a tensor
    with those strides reproduces the reference's values (plain ``torch.randn(B, C, T_y)`` does not); the
    C ABI takes the strides (``bv2_decode_in.nz_*stride``), so the tensor is handed over as it is."""
    return torch.randn_like(torch.empty(B, Ty, C, device=device, dtype=torch.float32).transpose(1, 2))


class _Node(nn.Module):
    """A bare namespace module: only there so that parameters carry the reference's dotted names."""

    def forward(self, *a, **k):  # pragma: no cover
        raise NotImplementedError("bert_vits2_amd exposes SynthesizerTrn.infer() only; sub-modules are parameter holders")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


class SynthesizerTrn(nn.Module):
    """Synthesizer (inference path) — see module docstring."""

    def __init__(self, n_vocab, spec_channels, segment_size, inter_channels, hidden_channels, filter_channels, n_heads,
                 n_layers, kernel_size, p_dropout, resblock, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_initial_channel, upsample_kernel_sizes, n_speakers=256, gin_channels=256,
                 use_sdp=True, n_flow_layer=4, n_layers_trans_flow=4, flow_share_parameter=False,
                 use_transformer_flow=True, **kwargs):
        super().__init__()
        self.hp = H.from_ctor(
            n_vocab, spec_channels, segment_size, inter_channels=inter_channels, hidden_channels=hidden_channels,
            filter_channels=filter_channels, n_heads=n_heads, n_layers=n_layers, kernel_size=kernel_size,
            p_dropout=p_dropout, resblock=resblock, resblock_kernel_sizes=list(resblock_kernel_sizes),
            resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes], upsample_rates=list(upsample_rates),
            upsample_initial_channel=upsample_initial_channel, upsample_kernel_sizes=list(upsample_kernel_sizes),
            n_speakers=n_speakers, gin_channels=gin_channels, use_sdp=use_sdp, n_flow_layer=n_flow_layer,
            n_layers_trans_flow=n_layers_trans_flow, flow_share_parameter=flow_share_parameter,
            use_transformer_flow=use_transformer_flow)
        self.hp.validate()
        # attributes the reference exposes and callers read
        self.n_vocab, self.spec_channels, self.segment_size = n_vocab, spec_channels, segment_size
        self.inter_channels, self.hidden_channels, self.filter_channels = inter_channels, hidden_channels, filter_channels
        self.n_heads, self.n_layers, self.kernel_size, self.p_dropout = n_heads, n_layers, kernel_size, p_dropout
        self.n_speakers, self.gin_channels = n_speakers, gin_channels
        self.use_sdp = use_sdp
        for key, shape in param_shapes(self.hp).items():
            # placeholders until a checkpoint is loaded: LayerNorm gamma / weight_norm g at 1 (the reference's own defaults,
            # modules.py:22-23), everything else 0.  infer() refuses to run on them (see repack).
            leaf = key.rsplit(".", 1)[-1]
            init = torch.ones if leaf in ("gamma", "weight_g") else torch.zeros
            self._register(key, init(shape, dtype=torch.float32))
        self._params_loaded = False
        self._lib = None
        self._handle = C.c_void_p()
        self._blob: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._taps: Dict[str, torch.Tensor] = {}
        self.generator_dtype = torch.float32
        self.flow_dtype = torch.float32
        self._graphs_on = False
        self._graphs_static = False
        self._ty_bucket = 32
        self.graph_stats = dict(captures=0, replays=0)
        self._graphs: Dict[tuple, dict] = {}
        self._cap_stream = None
        self._options: Dict[str, int] = {}

    # ------------------------------------------------------------------ parameter tree
    def _register(self, key: str, value: torch.Tensor) -> None:
        node = self
        parts = key.split(".")
        for p in parts[:-1]:
            if p not in node._modules:
                node.add_module(p, _Node())
            node = node._modules[p]
        node.register_parameter(parts[-1], nn.Parameter(value, requires_grad=False))

    def _invalidate_weights(self) -> None:
        """The packed device blob no longer matches the parameters (or is about to move): forget it everywhere — captured
        graphs baked its address in, and the C handle must not keep pointing at memory torch may free."""
        if getattr(self, "_graphs", None):
            self._drop_graphs()
        if getattr(self, "_lib", None) is not None and getattr(self, "_handle", None):
            self._lib.bv2_detach_weights(self._handle)
        self._blob = None

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        self._invalidate_weights()
        self._params_loaded = True
        if not strict:
            # the reference loads with strict=False and tolerates training-only keys (enc_q.*, sdp.post_*)
            own = set(k for k, _ in self.named_parameters())
            state_dict = {k: v for k, v in state_dict.items() if k in own}
        return super().load_state_dict(state_dict, strict=strict, **kw)

    def _apply(self, fn, *a, **k):
        if hasattr(self, "_blob"):
            self._invalidate_weights()
        self._ws = None
        return super()._apply(fn, *a, **k)

    def forward(self, *a, **k):
        raise NotImplementedError("training forward() is out of scope; this class accelerates infer() only")

    # ------------------------------------------------------------------ native handle / weights
    @property
    def device(self) -> torch.device:
        # a rank that only received the packed blob (sharding.distribute_weights) runs where the blob lives
        if self._blob is not None:
            return self._blob.device
        return next(self.parameters()).device

    def _ensure_handle(self):
        if self._lib is None:
            self._lib = L.load()
            cfg = L.make_config(self.hp)
            rc = self._lib.bv2_create(C.byref(cfg), C.byref(self._handle))
            if rc:
                raise RuntimeError("bv2_create failed: " + self._lib.bv2_last_error(None).decode())
        return self._lib

    def _check(self, rc: int, what: str):
        if rc:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.bv2_last_error(self._handle).decode()}")

    def __del__(self):
        try:
            self._drop_graphs()
            if self._lib is not None and self._handle:
                self._lib.bv2_destroy(self._handle)
        except Exception:
            pass

    def pack_host_blob(self) -> torch.Tensor:
        """Fold weight-norm / repack every tensor into the library's blob (host, uint8).  CPU-only: usable without a GPU."""
        lib = self._ensure_handle()
        for key, p in self.named_parameters():
            t = p.detach().to("cpu", torch.float32).contiguous()
            shape = (C.c_int64 * t.dim())(*t.shape)
            rc = lib.bv2_load_tensor(self._handle, key.encode(), C.c_void_p(t.data_ptr()), shape, t.dim(), L.F32)
            if rc < 0:                         # rc == 1: a key the inference path ignores (e.g. sdp.flows.1.*)
                self._check(rc, "bv2_load_tensor(" + key + ")")
        n = lib.bv2_packed_bytes(self._handle)
        blob = torch.empty(n, dtype=torch.uint8)
        self._check(lib.bv2_pack_weights(self._handle, C.c_void_p(blob.data_ptr()), n), "bv2_pack_weights")
        return blob

    def attach_blob(self, dev_blob: torch.Tensor) -> None:
        """Use an already packed blob resident on this GPU (e.g. received through an RCCL broadcast)."""
        lib = self._ensure_handle()
        assert dev_blob.is_cuda and dev_blob.dtype == torch.uint8 and dev_blob.is_contiguous()
        self._drop_graphs()
        self._check(lib.bv2_attach_weights(self._handle, C.c_void_p(dev_blob.data_ptr()), dev_blob.numel()),
                    "bv2_attach_weights")
        self._blob = dev_blob

    def repack(self) -> None:
        """(Re)build the packed device weights from the current parameters (call after editing parameters in place)."""
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("bert_vits2_amd.SynthesizerTrn.infer needs the model on a GPU (model.to('cuda')); "
                               "there is no CPU fallback")
        if not self._params_loaded:
            raise RuntimeError("no weights: this SynthesizerTrn never saw load_state_dict() / load_checkpoint() (a rank that only "
                               "received the packed blob must attach_blob() again after .to(); it has no parameters to repack)")
        with torch.cuda.device(dev):
            self.attach_blob(self.pack_host_blob().to(dev))

    def set_generator_dtype(self, dtype) -> None:
        """Arithmetic of the Generator (``dec``): ``torch.float32`` (default, exact fp32 MFMA) or ``torch.bfloat16``
        (bf16 weights + channels-last bf16 activations, fp32 accumulate — BASELINE config 3; the reference's counterpart
        is running ``dec`` under ``torch.autocast(bfloat16)``).  Encoder, durations and flow stay fp32 either way."""
        lib = self._ensure_handle()
        code = {torch.float32: L.F32, "fp32": L.F32, "f32": L.F32, torch.bfloat16: L.BF16, "bf16": L.BF16}.get(dtype)
        if code is None:
            raise ValueError("generator dtype must be torch.float32 or torch.bfloat16")
        self._check(lib.bv2_set_generator_dtype(self._handle, code), "bv2_set_generator_dtype")
        self._drop_graphs()
        self.generator_dtype = torch.bfloat16 if code == L.BF16 else torch.float32

    def set_flow_dtype(self, dtype) -> None:
        """Arithmetic of the transformer flow's Encoder convolutions (fused q/k/v, conv_o, FFN): ``torch.float32``
        (default) or ``torch.float16`` (fp16 weights / conv inputs / FFN hidden, fp32 accumulate — BASELINE config 5
        "fp16 flow + fp32 spline"; the reference's counterpart is ``flow`` under ``torch.autocast(float16)``); the attention
        core's QK^T / PV products take fp16 operands too.  LayerNorm, softmax, the residual stream and everything before the
        flow stay fp32: durations are unchanged."""
        lib = self._ensure_handle()
        code = {torch.float32: L.F32, "fp32": L.F32, "f32": L.F32, torch.float16: L.F16, "fp16": L.F16, "f16": L.F16}.get(dtype)
        if code is None:
            raise ValueError("flow dtype must be torch.float32 or torch.float16")
        self._check(lib.bv2_set_flow_dtype(self._handle, code), "bv2_set_flow_dtype")
        self._drop_graphs()
        self.flow_dtype = torch.float16 if code == L.F16 else torch.float32

    def _workspace(self, B: int, T: int, Ty: int) -> torch.Tensor:
        n = self._lib.bv2_workspace_bytes(self._handle, B, T, Ty)
        if n < 0:
            raise RuntimeError("bv2_workspace_bytes failed")
        if self._ws is None or self._ws.numel() < n or self._ws.device != self.device:
            self._drop_graphs()                    # captured graphs bake in the workspace address
            self._ws = torch.empty(int(n * 1.25), dtype=torch.uint8, device=self.device)
        return self._ws

    # ------------------------------------------------------------------ hipGraph replay of the two phases
    def enable_graphs(self, on: bool = True, static_io: bool = False, ty_bucket: int = 32) -> None:
        """Record each phase once per shape as a hipGraph (``bv2_graph_capture_*``) and replay it on later calls: one
        ``hipGraphLaunch`` per phase instead of ~230 kernel launches.  Default: inputs are copied into buffers the graph owns and
        outputs are returned as fresh tensors, so the call behaves exactly like the eager one.

        ``ty_bucket`` (frames): ``T_y = max(y_lengths)`` is data-dependent (reference commons.py:119-123, models.py:1058), so with real
        durations nearly every batch has a T_y of its own.  Phase B is therefore recorded once per BUCKET — T_y rounded up to a multiple of
        ``ty_bucket`` — and replayed for every T_y inside it: the flow masks frames past ``y_lengths`` as in any ragged batch, the Generator
        treats everything past the batch's longest utterance as zero padding (``bv2_decode_in.exact_lengths`` 2, or 1 with
        ``exact_lengths=True``), and the returned tensors are views cut back to the true T_y.  The results are those of the eager call up to
        fp32 summation order (a bucket may pick another split-K factor); ``ty_bucket=1`` keys graphs on the exact T_y (bit-identical to
        eager, one capture per distinct T_y).  ``graph_stats`` counts captures and replays.

        ``static_io=True`` (serving loops): no staging copies — a graph is recorded reading the caller's input tensors IN PLACE
        (when a tensor with another address shows up the shape's graph is re-recorded ONCE with input buffers of its own and the
        inputs are copied in from then on; up to 16 shapes are cached) and the returned tensors are the graph's own output buffers, valid until the next call with the same shapes.  Only the two noise draws still move:
        the CPU draw of models.py:248-251 is uploaded into the graph's buffer, the device draw of :1071 is made in place."""
        self._graphs_on = bool(on)
        self._graphs_static = bool(on and static_io)
        self._ty_bucket = max(1, int(ty_bucket))
        self.graph_stats = dict(captures=0, replays=0)
        self._drop_graphs()

    def _drop_graphs(self) -> None:
        graphs, self._graphs = getattr(self, "_graphs", {}), {}
        for g in graphs.values():
            if self._lib is not None and g.get("graph"):
                self._lib.bv2_graph_destroy(g["graph"])

    def _capture(self, fn, *args):
        """Run one capture call on a dedicated non-default stream (the legacy default stream cannot be captured)."""
        dev = self.device
        if self._cap_stream is None or self._cap_stream.device != dev:
            self._cap_stream = torch.cuda.Stream(device=dev)
        torch.cuda.current_stream(dev).synchronize()      # buffers the graph will touch are idle while it is recorded
        g = C.c_void_p()
        self._check(fn(self._handle, C.c_void_p(self._cap_stream.cuda_stream), *args, C.byref(g)), fn.__name__)
        return g

    def _graph_entry(self, key, build, ptrs=()):
        """Cached graph for ``key`` (shapes + scalars).  ``ptrs``: addresses of the tensors a static_io graph reads in place.  A
        graph recorded on other addresses is NOT re-recorded per call: the first pointer miss rebuilds the entry ONCE with input
        buffers the graph owns (``build(own_inputs=True)``) and from then on inputs are copied in — a caller that re-materialises
        its inputs every call (CPU tensors, the ``w_ceil=`` override, fresh uploads) costs one copy per call, not one capture."""
        ent = self._graphs.get(key)
        if ent is not None and not ent["own_inputs"] and ent["ptrs"] != ptrs:
            self._lib.bv2_graph_destroy(self._graphs.pop(key)["graph"])
            ent = None
            own = True
        else:
            own = not self._graphs_static
        if ent is None:
            if len(self._graphs) >= 16:                   # bounded cache: drop the least recently used shape
                old = next(iter(self._graphs))
                self._lib.bv2_graph_destroy(self._graphs.pop(old)["graph"])
            ent = build(own)
            ent["own_inputs"], ent["ptrs"] = own, ptrs
            self._graphs[key] = ent
            self.graph_stats["captures"] += 1
        else:
            self._graphs[key] = self._graphs.pop(key)     # most recently used last
            self.graph_stats["replays"] += 1
        return ent

    def set_tap(self, name: Optional[str], tensor: Optional[torch.Tensor] = None) -> None:
        """Debug: copy a named intermediate into ``tensor`` (fp32, CUDA) during the next calls."""
        lib = self._ensure_handle()
        if name is None:
            self._taps.clear()
            lib.bv2_set_tap(self._handle, None, None, 0)
            return
        if tensor is None:
            self._taps.pop(name, None)
            lib.bv2_set_tap(self._handle, name.encode(), None, 0)
            return
        self._taps[name] = tensor
        lib.bv2_set_tap(self._handle, name.encode(), C.c_void_p(tensor.data_ptr()), tensor.numel())

    # ------------------------------------------------------------------ the two phases
    @torch.no_grad()
    def encode_durations(self, x, x_lengths, sid, tone, language, bert, ja_bert, en_bert, noise_w, noise_scale_w=0.8,
                         sdp_ratio=0.0, length_scale=1.0, bert_index=None) -> Dict[str, torch.Tensor]:
        """Phase A = reference models.py:1045-1057.  ``noise_w`` [B,2,T] is the draw of models.py:248-251.

        ``bert_index`` (optional, a 3-tuple aligned with bert / ja_bert / en_bert; entries may be None): a feature with an index is
        WORD-level — [B, 1024, S] as the BERT model emitted it, still on the device — and symbol t reads column index[b, t]; the
        word2ph repeat of reference text/chinese_bert.py:48-58 happens as a gather inside the TextEncoder front
        (``bert_features.word_level_feature`` builds the pair)."""
        if self._blob is None:
            self.repack()
        dev = self.device
        B, T = x.shape
        i64 = lambda t: t.to(dev, torch.int64).contiguous()
        f32 = lambda t: t.to(dev, torch.float32).contiguous()
        x, x_lengths, sid, tone, language = i64(x), i64(x_lengths), i64(sid), i64(tone), i64(language)
        bert, ja_bert, en_bert, noise_w = f32(bert), f32(ja_bert), f32(en_bert), f32(noise_w)
        bidx = [None, None, None] if bert_index is None else [None if t is None else t.to(dev, torch.int32).contiguous() for t in bert_index]
        for feat, ix in zip((bert, ja_bert, en_bert), bidx):
            want = (B, H.BERT_DIM, T) if ix is None else (B, H.BERT_DIM, feat.shape[2])
            if tuple(feat.shape) != want or (ix is not None and (tuple(ix.shape) != (B, T) or not 1 <= feat.shape[2] <= T)):
                raise ValueError(f"bert features must be [B,{H.BERT_DIM},T] (reference infer.py:124), or [B,{H.BERT_DIM},S<=T] with a [B,T] index")
        cols = [0 if ix is None else int(f.shape[2]) for f, ix in zip((bert, ja_bert, en_bert), bidx)]

        def with_index(ein, ptr_of):
            for i in range(3):
                ein.bert_index[i] = None if bidx[i] is None else ptr_of(i)
                ein.bert_cols[i] = cols[i]
            return ein
        if noise_w.shape != (B, 2, T):
            raise ValueError("noise_w must be [B,2,T]")
        hp = self.hp
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        okeys = ("g", "x", "m_p", "logs_p", "x_mask", "logw_sdp", "logw_dp", "logw", "w_ceil", "y_lengths")
        mk_out = lambda: dict(g=e(B, hp.gin_channels), x=e(B, hp.hidden_channels, T), m_p=e(B, hp.inter_channels, T),
                              logs_p=e(B, hp.inter_channels, T), x_mask=e(B, T), logw_sdp=e(B, T), logw_dp=e(B, T),
                              logw=e(B, T), w_ceil=e(B, T), y_lengths=torch.empty(B, dtype=torch.int64, device=dev))
        if self._graphs_on and not self._taps:
            ws = self._workspace(B, T, 1)
            ins = dict(x=x, x_lengths=x_lengths, sid=sid, tone=tone, language=language, bert=bert, ja_bert=ja_bert,
                       en_bert=en_bert, noise_w=noise_w)
            for i in range(3):
                if bidx[i] is not None:
                    ins[f"bert_index{i}"] = bidx[i]

            static = self._graphs_static

            def build(own_inputs):
                staged = tuple(ins) if own_inputs else ("noise_w",)     # static_io: everything but the noise is read in place
                sin = {k: (torch.empty_like(v) if k in staged else v) for k, v in ins.items()}
                sout = mk_out()
                ein = L.EncodeIn(B, T, *[_ptr(sin[k]) for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert",
                                                                "en_bert", "noise_w")],
                                 float(noise_scale_w), float(sdp_ratio), float(length_scale))
                with_index(ein, lambda i: sin[f"bert_index{i}"].data_ptr())
                eout = L.EncodeOut(*[_ptr(sout[k]) for k in okeys])
                with torch.cuda.device(dev):
                    g = self._capture(self._lib.bv2_graph_capture_encode, C.byref(ein), C.byref(eout),
                                      C.c_void_p(ws.data_ptr()), ws.numel())
                return dict(graph=g, sin=sin, sout=sout, staged=staged)

            ptrs = tuple(ins[k].data_ptr() for k in ins if k != "noise_w") if static else ()
            ent = self._graph_entry(("A", B, T, float(noise_scale_w), float(sdp_ratio), float(length_scale), tuple(cols),
                                     tuple(sorted(ins))), build, ptrs)
            for k in ent["staged"]:
                ent["sin"][k].copy_(ins[k], non_blocking=True)
            with torch.cuda.device(dev):
                if self._lib.bv2_graph_launch(ent["graph"], C.c_void_p(torch.cuda.current_stream().cuda_stream)):
                    raise RuntimeError("bv2_graph_launch (encode) failed")
            if static:
                return dict(ent["sout"])
            return {k: v.clone() for k, v in ent["sout"].items()}
        out = mk_out()
        ein = L.EncodeIn(B, T, _ptr(x), _ptr(x_lengths), _ptr(sid), _ptr(tone), _ptr(language), _ptr(bert), _ptr(ja_bert),
                         _ptr(en_bert), _ptr(noise_w), float(noise_scale_w), float(sdp_ratio), float(length_scale))
        with_index(ein, lambda i: bidx[i].data_ptr())
        eout = L.EncodeOut(*[_ptr(out[k]) for k in okeys])
        ws = self._workspace(B, T, 1)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._check(self._lib.bv2_encode_durations(self._handle, stream, C.byref(ein), C.byref(eout),
                                                       C.c_void_p(ws.data_ptr()), ws.numel()), "bv2_encode_durations")
        return out

    @torch.no_grad()
    def decode(self, enc: Dict[str, torch.Tensor], noise_z: torch.Tensor, Ty: int, noise_scale=0.667, max_len=None,
               want_attn: bool = True, exact_lengths: bool = False, ty_bucket: Optional[int] = None) -> Dict[str, torch.Tensor]:
        """Phase B = reference models.py:1058-1073.  ``noise_z`` [B,inter,>=Ty] replaces randn_like at :1071.
        ``ty_bucket``: run at T_y rounded up to a multiple of it and cut the outputs back (default: the ``enable_graphs`` bucket when a
        graph is replayed, none on the eager path; tests pass it to the eager path to compare like with like)."""
        if self._blob is None:
            self.repack()
        dev = self.device
        hp = self.hp
        B, _, T = enc["x"].shape
        Ci = hp.inter_channels
        draw_z = noise_z is None                        # infer(): draw #2 straight into the graph's buffer (static_io)
        if not draw_z:
            assert noise_z.is_cuda and noise_z.dtype == torch.float32 and noise_z.shape[2] >= Ty and noise_z.shape[1] == Ci
        L_dec = Ty if (max_len is None or max_len <= 0 or max_len >= Ty) else int(max_len)
        S = L_dec * hp.total_upsample
        okeys = ("o", "attn", "y_mask", "z", "z_p", "m_p", "logs_p")
        # ---- T_y bucket: everything below runs at Ty (the bucket); Ty_true / S_true cut the results back
        if ty_bucket is None:
            ty_bucket = self._ty_bucket if (self._graphs_on and not self._taps) else 1
        Ty_true, S_true, L_true = Ty, S, L_dec
        mode = int(bool(exact_lengths))
        if ty_bucket > 1:
            Ty = (Ty + ty_bucket - 1) // ty_bucket * ty_bucket
            if L_dec == Ty_true:                        # no max_len cut: the Generator runs over the bucket, capped on the device
                L_dec = Ty
                mode = 1 if exact_lengths else 2
            S = L_dec * hp.total_upsample
        bucketed = Ty != Ty_true

        def cut(out):
            if not bucketed:
                return out
            r = dict(out)
            r["o"] = out["o"][:, :, :S_true]
            for k in ("z", "z_p", "m_p", "logs_p", "y_mask"):
                r[k] = None if out[k] is None else out[k][:, :, :Ty_true]
            r["attn"] = None if out["attn"] is None else out["attn"][:, :, :Ty_true]
            return r

        def mk_out():
            # ONE allocation carved into the six secondary outputs (+ one for the waveform): this runs between the reference's host sync and the first launch of
            # phase B, i.e. with the GPU idle — seven caching-allocator calls there cost ~25 us of a 4.5 ms step at batch 1
            shapes = dict(o=(B, 1, S), attn=(B, 1, Ty, T) if want_attn else None, y_mask=(B, 1, Ty), z=(B, Ci, Ty), z_p=(B, Ci, Ty),
                          m_p=(B, Ci, Ty), logs_p=(B, Ci, Ty))
            # The waveform gets its OWN allocation: a caller that keeps only `o` (the serving case) must not pin attn and the four
            # [B,Ci,Ty] latents with it.
            sizes = {k: (0 if sh is None or k == "o" else (math.prod(sh) + 63) // 64 * 64) for k, sh in shapes.items()}   # 256-byte aligned views
            flat = torch.empty(sum(sizes.values()), dtype=torch.float32, device=dev)
            out, off = {"o": torch.empty(shapes["o"], dtype=torch.float32, device=dev)}, 0
            for k, sh in shapes.items():
                if k == "o":
                    continue
                out[k] = None if sh is None else flat[off:off + math.prod(sh)].view(*sh)
                off += sizes[k]
            return out
        if self._graphs_on and not self._taps:
            ws = self._workspace(B, T, Ty)
            ikeys = ("m_p", "logs_p", "x_mask", "w_ceil", "y_lengths", "g")

            static = self._graphs_static

            def build(own_inputs):
                sin = {k: (torch.empty_like(enc[k]) if own_inputs else enc[k]) for k in ikeys}
                # the reference's strides for the prior noise (draw_noise_z): an in-place normal_() on it IS randn_like(m_p)
                sin["noise_z"] = torch.zeros(B, Ty, Ci, dtype=torch.float32, device=dev).transpose(1, 2)   # zeros: a bucket's tail is never drawn
                sout = mk_out()
                din = L.DecodeIn(B, T, int(Ty), int(L_dec), *[_ptr(sin[k]) for k in ikeys], _ptr(sin["noise_z"]),
                                 sin["noise_z"].stride(0), sin["noise_z"].stride(1), sin["noise_z"].stride(2),
                                 float(noise_scale), mode)
                dout = L.DecodeOut(*[_ptr(sout[k]) for k in okeys])
                with torch.cuda.device(dev):
                    g = self._capture(self._lib.bv2_graph_capture_decode, C.byref(din), C.byref(dout),
                                      C.c_void_p(ws.data_ptr()), ws.numel())
                return dict(graph=g, sin=sin, sout=sout)

            ptrs = tuple(enc[k].data_ptr() for k in ikeys) if static else ()
            ent = self._graph_entry(("B", B, T, int(Ty), int(L_dec), bool(want_attn), float(noise_scale), mode), build, ptrs)
            if ent["own_inputs"]:
                for k in ikeys:
                    ent["sin"][k].copy_(enc[k])
            if draw_z and not bucketed:
                ent["sin"]["noise_z"].normal_()
            elif draw_z:                                # the reference draws exactly [B, C, T_y] in its memory order (draw_noise_z): keep the RNG contract
                ent["sin"]["noise_z"][:, :, :Ty_true].copy_(draw_noise_z(B, Ci, Ty_true, dev))
            else:
                ent["sin"]["noise_z"][:, :, :Ty_true].copy_(noise_z[:, :, :Ty_true])
            with torch.cuda.device(dev):
                if self._lib.bv2_graph_launch(ent["graph"], C.c_void_p(torch.cuda.current_stream().cuda_stream)):
                    raise RuntimeError("bv2_graph_launch (decode) failed")
            if static:
                return cut(dict(ent["sout"]))
            return {k: (None if v is None else v.clone()) for k, v in cut(ent["sout"]).items()}
        if draw_z:
            noise_z = draw_noise_z(B, Ci, Ty_true, dev)
        if bucketed:                                    # eager run at a bucket (tests): zero tail, reference-strided like the graph's buffer
            nzb = torch.zeros(B, Ty, Ci, dtype=torch.float32, device=dev).transpose(1, 2)
            nzb[:, :, :Ty_true].copy_(noise_z[:, :, :Ty_true])
            noise_z = nzb
        out = mk_out()
        din = L.DecodeIn(B, T, int(Ty), int(L_dec), _ptr(enc["m_p"]), _ptr(enc["logs_p"]), _ptr(enc["x_mask"]),
                         _ptr(enc["w_ceil"]), _ptr(enc["y_lengths"]), _ptr(enc["g"]), _ptr(noise_z),
                         noise_z.stride(0), noise_z.stride(1), noise_z.stride(2), float(noise_scale), mode)
        dout = L.DecodeOut(*[_ptr(out[k]) for k in okeys])
        ws = self._workspace(B, T, Ty)
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._check(self._lib.bv2_decode(self._handle, stream, C.byref(din), C.byref(dout), C.c_void_p(ws.data_ptr()),
                                             ws.numel()), "bv2_decode")
        return cut(out)

    # ------------------------------------------------------------------ the reference entry point
    @torch.no_grad()
    def infer(self, x, x_lengths, sid, tone, language, bert, ja_bert, en_bert, noise_scale=0.667, length_scale=1,
              noise_scale_w=0.8, max_len=None, sdp_ratio=0, y=None, *, noise_w=None, noise_z=None, w_ceil=None,
              want_attn=True, exact_lengths=False, bert_index=None, ty_bucket=None):
        """reference models.py:1026-1074.  Keyword-only extras (not in the reference): ``noise_w`` [B,2,T] and
        ``noise_z`` [B,inter,>=T_y] inject the two N(0,1) draws (parity tests; the reference's ONNX export externalises
        them the same way), ``w_ceil`` substitutes the durations, ``want_attn=False`` skips materialising the path,
        ``exact_lengths=True`` makes every utterance of a ragged batch come out exactly as if it had been run alone (the
        reference's unmasked decoder lets the padding bleed into an utterance's last ~40 ms; bv2.h ``exact_lengths``),
        ``bert_index`` hands BERT features over at word level (see ``encode_durations``), ``ty_bucket`` runs phase B at T_y rounded up to
        a multiple of it (default: ``enable_graphs``' bucket when graphs are on, exact T_y otherwise; see ``enable_graphs``)."""
        if self.device.type != "cuda":
            raise RuntimeError("bert_vits2_amd.SynthesizerTrn.infer needs a GPU: no CPU fallback exists by design")
        dev = self.device
        B, T = x.shape
        if noise_w is None:
            noise_w = draw_noise_w(B, T, dev)
        enc = self.encode_durations(x, x_lengths, sid, tone, language, bert, ja_bert, en_bert, noise_w,
                                    noise_scale_w=noise_scale_w, sdp_ratio=sdp_ratio, length_scale=length_scale,
                                    bert_index=bert_index)
        if w_ceil is not None:
            wc = w_ceil.to(dev, torch.float32).reshape(B, T).contiguous()
            enc["w_ceil"] = wc
            enc["y_lengths"] = torch.clamp_min(wc.sum(1), 1).long()
        # the reference's one host sync (commons.py:120-122).  One utterance: no reduction kernel in front of the copy.
        Ty = int(enc["y_lengths"].item()) if B == 1 else int(enc["y_lengths"].max().item())
        if noise_z is not None:                        # None: decode() draws it (in place in the graph's buffer when replaying)
            noise_z = noise_z.to(dev, torch.float32)
        dec = self.decode(enc, noise_z, Ty, noise_scale=noise_scale, max_len=max_len, want_attn=want_attn,
                          exact_lengths=exact_lengths, ty_bucket=ty_bucket)
        self.last_encode = enc
        return dec["o"], dec["attn"], dec["y_mask"], (dec["z"], dec["z_p"], dec["m_p"], dec["logs_p"])

    # ------------------------------------------------------------------ single stages (reference ONNX seams)
    # onnx_modules/V230/models_onnx.py:896-1063 cuts infer() into emb_g / enc_p / sdp / dp / flow / dec; the six methods below
    # are those graphs (same tensor names, see onnx_api.StageSession for the consumer-side glue).
    def _stage_call(self, fn, B, T_or_Ty, *ptr_args, what):
        ws = self._workspace(B, max(int(T_or_Ty), 1), max(int(T_or_Ty), 1))
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._check(fn(self._handle, stream, *ptr_args, C.c_void_p(ws.data_ptr()), ws.numel()), what)

    def _f32(self, t, *shape):
        t = t.to(self.device, torch.float32)
        return (t.reshape(*shape) if shape else t).contiguous()

    def _i64(self, t):
        return t.to(self.device, torch.int64).contiguous()

    @torch.no_grad()
    def stage_emb_g(self, sid: torch.Tensor) -> torch.Tensor:
        """``emb_g.run({"sid"})`` -> g [B, gin]."""
        if self._blob is None:
            self.repack()
        sid = self._i64(sid).reshape(-1)
        B = sid.shape[0]
        g = torch.empty(B, self.hp.gin_channels, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            self._check(self._lib.bv2_stage_emb_g(self._handle, stream, B, _ptr(sid), _ptr(g)), "bv2_stage_emb_g")
        return g

    @torch.no_grad()
    def stage_enc_p(self, x, t, language, bert_0, bert_1, bert_2, g, x_lengths=None):
        """``enc.run({"x","t","language","bert_0","bert_1","bert_2","g"})`` -> (xout, m_p, logs_p, x_mask [B,1,T])."""
        if self._blob is None:
            self.repack()
        x, t, language = self._i64(x), self._i64(t), self._i64(language)
        B, T = x.shape
        b0, b1, b2 = (self._f32(v, B, H.BERT_DIM, T) for v in (bert_0, bert_1, bert_2))
        g = self._f32(g, B, -1)
        xl = None if x_lengths is None else self._i64(x_lengths)
        hp, dev = self.hp, self.device
        e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        xout, m_p, logs_p, x_mask = e(B, hp.hidden_channels, T), e(B, hp.inter_channels, T), e(B, hp.inter_channels, T), e(B, 1, T)
        self._stage_call(self._lib.bv2_stage_enc_p, B, T, B, T, _ptr(x), _ptr(t), _ptr(language), _ptr(b0), _ptr(b1), _ptr(b2),
                         _ptr(g), _ptr(xl), _ptr(xout), _ptr(m_p), _ptr(logs_p), _ptr(x_mask), what="bv2_stage_enc_p")
        return xout, m_p, logs_p, x_mask

    @torch.no_grad()
    def stage_sdp(self, x, x_mask, zin, g) -> torch.Tensor:
        """``sdp.run({"x","x_mask","zin","g"})`` -> logw [B,1,T]; ``zin`` [B,2,T] is the already scaled noise."""
        if self._blob is None:
            self.repack()
        B, _, T = x.shape
        x, x_mask, zin, g = self._f32(x), self._f32(x_mask, B, T), self._f32(zin, B, 2, T), self._f32(g, B, -1)
        logw = torch.empty(B, 1, T, dtype=torch.float32, device=self.device)
        self._stage_call(self._lib.bv2_stage_sdp, B, T, B, T, _ptr(x), _ptr(x_mask), _ptr(zin), _ptr(g), _ptr(logw),
                         what="bv2_stage_sdp")
        return logw

    @torch.no_grad()
    def stage_dp(self, x, x_mask, g) -> torch.Tensor:
        """``dp.run({"x","x_mask","g"})`` -> logw [B,1,T]."""
        if self._blob is None:
            self.repack()
        B, _, T = x.shape
        x, x_mask, g = self._f32(x), self._f32(x_mask, B, T), self._f32(g, B, -1)
        logw = torch.empty(B, 1, T, dtype=torch.float32, device=self.device)
        self._stage_call(self._lib.bv2_stage_dp, B, T, B, T, _ptr(x), _ptr(x_mask), _ptr(g), _ptr(logw), what="bv2_stage_dp")
        return logw

    @torch.no_grad()
    def stage_flow(self, z_p: torch.Tensor, y_lengths: Optional[torch.Tensor], g: torch.Tensor,
                   y_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``flow.run({"z_p","y_mask","g"})`` -> z.  The frame mask is given as ``y_lengths`` [B] or as ``y_mask`` [B,1,Ty]."""
        if self._blob is None:
            self.repack()
        B, Ci, Ty = z_p.shape
        z_p = self._f32(z_p)
        z = torch.empty_like(z_p)
        yl = None if y_lengths is None else self._i64(y_lengths)
        ym = None if y_mask is None else self._f32(y_mask, B, Ty)
        if (yl is None) == (ym is None):
            raise ValueError("stage_flow needs exactly one of y_lengths / y_mask")
        g = self._f32(g, B, -1)
        self._stage_call(self._lib.bv2_stage_flow, B, Ty, B, Ty, _ptr(z_p), _ptr(yl), _ptr(ym), _ptr(g), _ptr(z),
                         what="bv2_stage_flow")
        return z

    @torch.no_grad()
    def stage_generator(self, z: torch.Tensor, y_lengths: Optional[torch.Tensor], g: torch.Tensor,
                        L_frames: Optional[int] = None):
        """``dec.run({"z_in","g"})`` -> o [B,1,L*hop].  With ``y_lengths`` the input is (z*y_mask)[:, :, :L] (models.py:1073);
        with ``None`` z is taken as it is (the exported graph)."""
        if self._blob is None:
            self.repack()
        B, Ci, Ty = z.shape
        Lf = Ty if L_frames is None else int(L_frames)
        z = self._f32(z)
        yl = None if y_lengths is None else self._i64(y_lengths)
        g = self._f32(g, B, -1)
        o = torch.empty(B, 1, Lf * self.hp.total_upsample, dtype=torch.float32, device=self.device)
        self._stage_call(self._lib.bv2_stage_generator, B, Ty, B, Ty, Lf, _ptr(z), _ptr(yl), _ptr(g), _ptr(o),
                         what="bv2_stage_generator")
        return o

    def set_option(self, key: str, value: int) -> None:
        """Kernel-selection switches of ``bv2_set_option`` ("fused_resblock", "fused_dds"); tests only."""
        self._ensure_handle()
        self._check(self._lib.bv2_set_option(self._handle, key.encode(), int(value)), "bv2_set_option")
        self._options[key] = int(value)
        self._drop_graphs()

    # ------------------------------------------------------------------ measurement
    def profile(self, on=1):
        """0 off, 1 time every MFMA kernel launch with HIP events, 2 only the Generator's launches."""
        self._ensure_handle()
        self._check(self._lib.bv2_profile_enable(self._handle, int(on)), "bv2_profile_enable")
        self._lib.bv2_profile_reset(self._handle)

    def profile_pause(self):
        """Stop recording events but keep what was recorded (the pool holds 8192 launches)."""
        self._check(self._lib.bv2_profile_enable(self._handle, 0), "bv2_profile_enable")

    def profile_report(self):
        rows = (L.ProfileRow * 256)()
        n = self._lib.bv2_profile_report(self._handle, rows, 256)
        out = []
        for i in range(max(n, 0)):
            r = rows[i]
            out.append(dict(name=r.name.decode(), launches=r.launches, total_ms=r.total_ms, flops=r.flops, bytes=r.bytes))
        self._lib.bv2_profile_reset(self._handle)
        return out


def from_hparams(hp: H.HParams) -> SynthesizerTrn:
    return SynthesizerTrn(
        hp.n_vocab, hp.spec_channels, hp.segment_size, inter_channels=hp.inter_channels, hidden_channels=hp.hidden_channels,
        filter_channels=hp.filter_channels, n_heads=hp.n_heads, n_layers=hp.n_layers, kernel_size=hp.kernel_size,
        p_dropout=hp.p_dropout, resblock=hp.resblock, resblock_kernel_sizes=hp.resblock_kernel_sizes,
        resblock_dilation_sizes=hp.resblock_dilation_sizes, upsample_rates=hp.upsample_rates,
        upsample_initial_channel=hp.upsample_initial_channel, upsample_kernel_sizes=hp.upsample_kernel_sizes,
        n_speakers=hp.n_speakers, gin_channels=hp.gin_channels, use_sdp=hp.use_sdp, n_flow_layer=hp.n_flow_layer,
        n_layers_trans_flow=hp.n_layers_trans_flow, flow_share_parameter=hp.flow_share_parameter,
        use_transformer_flow=hp.use_transformer_flow)
