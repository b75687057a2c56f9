"""Batched serving glue around ``SynthesizerTrn.infer``.

The reference synthesises the pieces of a request one at a time at batch 1 — ``infer.infer`` (infer.py:268-332) per
sentence / per ``|``-separated piece (webui.py:66-135, hiyoriUI.py:319-349), ``torch.cuda.empty_cache()`` after each —
then converts to 16-bit on the host.  Nothing in ``infer()`` couples batch elements, so here the
pieces of one or many requests are padded into length-bucketed batches, run through ONE ``infer()`` per bucket, cut back
to their own lengths on the device (with ``exact_lengths`` every utterance gets exactly the audio it gets alone, although
the reference's decoder is unmasked), and (optionally) converted to 16-bit PCM on the device (``bv2_pcm16``) before the
single device->host copy.  Multi-GPU: ``sharding.shard_indices`` picks this rank's utterances first.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import hparams as H


@dataclass
class Utterance:
    """One piece as reference ``infer.get_text`` (infer.py:107-152) produces it: 1-D ids and ``[1024, T]`` features."""
    phones: torch.Tensor
    tones: torch.Tensor
    lang_ids: torch.Tensor
    bert: torch.Tensor
    ja_bert: torch.Tensor
    en_bert: torch.Tensor
    sid: int = 0

    def __post_init__(self):
        T = int(self.phones.shape[0])
        if self.tones.shape != (T,) or self.lang_ids.shape != (T,):
            raise ValueError("phones / tones / lang_ids must be 1-D with the same length")
        for f in (self.bert, self.ja_bert, self.en_bert):
            if tuple(f.shape) != (H.BERT_DIM, T):                     # the reference asserts the same (infer.py:124)
                raise ValueError(f"bert features must be [{H.BERT_DIM}, {T}], got {tuple(f.shape)}")

    @property
    def length(self) -> int:
        return int(self.phones.shape[0])


def plan_batches(lengths: Sequence[int], max_batch: int = 32, max_pad_ratio: float = 1.25) -> List[List[int]]:
    """Length-bucketed batches: utterances sorted by length, a batch is closed when it is full or when its longest
    member would exceed ``max_pad_ratio`` x its shortest (padding is wasted work: every kernel runs over B x T_max)."""
    if max_batch < 1:
        raise ValueError("max_batch must be >= 1")
    order = sorted(range(len(lengths)), key=lambda i: (int(lengths[i]), i))
    batches, cur = [], []
    for i in order:
        if cur and (len(cur) >= max_batch or lengths[i] > max_pad_ratio * lengths[cur[0]]):
            batches.append(cur)
            cur = []
        cur.append(i)
    if cur:
        batches.append(cur)
    return batches


def collate(utts: Sequence[Utterance], device) -> dict:
    """Zero-pad to the longest utterance of the batch (padded symbols are masked out by ``x_lengths``)."""
    B, T = len(utts), max(u.length for u in utts)
    out = dict(x=torch.zeros(B, T, dtype=torch.int64), tone=torch.zeros(B, T, dtype=torch.int64),
               language=torch.zeros(B, T, dtype=torch.int64), x_lengths=torch.tensor([u.length for u in utts], dtype=torch.int64),
               sid=torch.tensor([u.sid for u in utts], dtype=torch.int64),
               bert=torch.zeros(B, H.BERT_DIM, T), ja_bert=torch.zeros(B, H.BERT_DIM, T), en_bert=torch.zeros(B, H.BERT_DIM, T))
    for i, u in enumerate(utts):
        n = u.length
        out["x"][i, :n], out["tone"][i, :n], out["language"][i, :n] = u.phones, u.tones, u.lang_ids
        out["bert"][i, :, :n], out["ja_bert"][i, :, :n], out["en_bert"][i, :, :n] = u.bert, u.ja_bert, u.en_bert
    return {k: v.to(device, non_blocking=True) for k, v in out.items()}


def pcm16(model, wave: torch.Tensor, y_lengths: torch.Tensor) -> torch.Tensor:
    """Device-side 16-bit conversion of ``wave`` [B,1,S] (peak-normalised per utterance over its valid samples, the
    semantics of gradio ``convert_to_16_bit_wav`` used by reference webui.py:86) -> int16 [B,S]."""
    lib = model._ensure_handle()
    B, _, S = wave.shape
    wave = wave.contiguous()
    out = torch.empty(B, S, dtype=torch.int16, device=wave.device)
    peak = torch.empty(B, dtype=torch.int32, device=wave.device)
    yl = y_lengths.to(wave.device, torch.int64).contiguous()
    with torch.cuda.device(wave.device):
        rc = lib.bv2_pcm16(C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(wave.data_ptr()), S,
                           C.c_void_p(yl.data_ptr()), model.hp.total_upsample, B, S, C.c_void_p(out.data_ptr()), S,
                           C.c_void_p(peak.data_ptr()))
    if rc:
        raise RuntimeError(f"bv2_pcm16 failed ({rc})")
    return out


def replicas(model, n: int) -> list:
    """``n`` shim instances that share ``model``'s packed weight blob (one copy in HBM), each with its own C handle, workspace and
    HIP stream: the unit of request-level concurrency.  A request's phase A and flow are chains of small kernels that leave most CUs
    idle; with a second request in flight on another stream its Generator fills them (batch-1 requests: 938 -> 1 265 audio-s/s with
    two in flight, 1 510 with four on MI355X, ``bench.py``'s ``config2_*_requests_in_flight``).  Cached on the model."""
    from . import models as _models
    if model.device.type != "cuda":
        raise RuntimeError("bert_vits2_amd.serving needs the model on a GPU: there is no CPU fallback")
    if model._blob is None:
        model.repack()
    reps = getattr(model, "_serving_replicas", None)
    if reps is None or getattr(model, "_serving_replicas_device", None) != model.device:
        # the cache belongs to ONE device: after model.to(another GPU) the old streams (and the replicas' handles) are on the wrong one
        reps = [(model, torch.cuda.Stream(model.device))]
    while len(reps) < n:
        with torch.device("meta"):                       # a replica only holds a handle: no CPU parameter set is materialised for it
            m = _models.from_hparams(model.hp)
        m.attach_blob(model._blob)
        reps.append((m, torch.cuda.Stream(model.device)))
    for m, _ in reps[1:]:                                # replicas follow the weights, precision switches and options of the model they serve
        if m._blob is not model._blob:
            m.attach_blob(model._blob)
        # the setters drop captured graphs: only call them on a real change (both flow variants have an fp16 form)
        if m.generator_dtype != model.generator_dtype:
            m.set_generator_dtype(model.generator_dtype)
        if m.flow_dtype != model.flow_dtype:
            m.set_flow_dtype(model.flow_dtype)
        for key, val in getattr(model, "_options", {}).items():
            if getattr(m, "_options", {}).get(key) != val:
                m.set_option(key, val)
        if (m._graphs_on, m._graphs_static) != (model._graphs_on, model._graphs_static):
            m.enable_graphs(model._graphs_on, static_io=model._graphs_static)
    model._serving_replicas, model._serving_replicas_device = reps, model.device
    return reps[:n]


@torch.no_grad()
def synthesize(model, utts: Sequence[Utterance], *, sdp_ratio=0.5, noise_scale=0.6, noise_scale_w=0.9, length_scale=1.0,
               max_batch: int = 32, max_pad_ratio: float = 1.25, as_pcm16: bool = False,
               noise: Optional[Sequence] = None, requests_in_flight: int = 1) -> List[np.ndarray]:
    """Synthesise every utterance; returns one 1-D array per utterance in input order (float32 like reference
    infer.py:315-319, or int16 with ``as_pcm16``).  Defaults are the reference web UI's (webui.py:443-454).
    ``noise`` (tests): per utterance a pair ``(noise_w [2,T], noise_z [inter, >= T_y])`` to inject instead of drawing.
    ``requests_in_flight`` > 1: the buckets are dealt round-robin to that many ``replicas`` (own handle + HIP stream, shared
    weights), so one bucket's small-kernel phases overlap another's Generator; results do not depend on it."""
    if model.device.type != "cuda":
        raise RuntimeError("bert_vits2_amd.serving needs the model on a GPU: there is no CPU fallback")
    dev = model.device
    hop = model.hp.total_upsample
    results: List[Optional[np.ndarray]] = [None] * len(utts)
    pending = []
    lanes = replicas(model, requests_in_flight) if requests_in_flight > 1 else [(model, None)]
    if requests_in_flight > 1:
        torch.cuda.current_stream(dev).synchronize()       # inputs prepared on the caller's stream are visible to the lanes
    for bi, idx in enumerate(plan_batches([u.length for u in utts], max_batch, max_pad_ratio)):
        lane, lane_stream = lanes[bi % len(lanes)]
        with torch.cuda.stream(lane_stream) if lane_stream is not None else contextlib.nullcontext():
            _run_bucket(lane, utts, idx, dev, noise, pending, as_pcm16, sdp_ratio, noise_scale, noise_scale_w, length_scale)
    for idx, host, y_len, ev in pending:
        ev.synchronize()
        for r, i in enumerate(idx):
            results[i] = host[r, :int(y_len[r]) * hop].numpy().copy()
    return results


def _run_bucket(model, utts, idx, dev, noise, pending, as_pcm16, sdp_ratio, noise_scale, noise_scale_w, length_scale):
    """One bucket on the CURRENT stream: collate, infer (exact lengths), optional PCM16, async D2H into pinned memory."""
    group = [utts[i] for i in idx]
    batch = collate(group, dev)
    kw = {}
    if noise is not None:
        T = batch["x"].shape[1]
        Tz = max(int(noise[i][1].shape[1]) for i in idx)
        nw = torch.zeros(len(idx), 2, T)
        nz = torch.zeros(len(idx), model.hp.inter_channels, Tz)
        for r, i in enumerate(idx):
            nw[r, :, :noise[i][0].shape[1]] = noise[i][0]
            nz[r, :, :noise[i][1].shape[1]] = noise[i][1]
        kw = dict(noise_w=nw.to(dev), noise_z=nz.to(dev))
    o, _attn, y_mask, _ = model.infer(batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"],
                                      batch["bert"], batch["ja_bert"], batch["en_bert"], sdp_ratio=sdp_ratio,
                                      noise_scale=noise_scale, noise_scale_w=noise_scale_w, length_scale=length_scale,
                                      want_attn=False, exact_lengths=True, **kw)
    y_len = model.last_encode["y_lengths"]             # int64 [B], already on the device (phase A output)
    audio = pcm16(model, o, y_len) if as_pcm16 else o[:, 0]
    # one async D2H per bucket into pinned memory (audio AND lengths): nothing here blocks the host, so the next bucket's
    # kernels are enqueued while this copy runs; the drain loop below waits on the bucket's event
    host = torch.empty(audio.shape, dtype=audio.dtype, pin_memory=True)
    host.copy_(audio, non_blocking=True)
    host_len = torch.empty(y_len.shape, dtype=torch.int64, pin_memory=True)
    host_len.copy_(y_len, non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    pending.append((idx, host, host_len, ev))
