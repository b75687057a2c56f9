"""The reference's ONNX stage cut of ``infer()`` on the HIP path.

The reference exports six graphs — ``emb_g``, ``enc_p``, ``sdp``, ``dp``, ``flow``, ``dec``
(onnx_modules/V230/models_onnx.py:896-1063) — and runs them from
``onnx_modules/V230_OnnxInference/__init__.py:36-126`` (``OnnxInferenceSession``): numpy glue between
``InferenceSession.run(None, {name: array})`` calls.  This module offers the same two layers on ``libbv2.so``:

* ``StageRunner`` — one exported graph behind the ``onnxruntime.InferenceSession.run(output_names, feeds)`` protocol, with the
  reference's tensor names (``sid`` / ``x, t, language, bert_0, bert_1, bert_2, g`` / ``x, x_mask, zin, g`` / ``x, x_mask, g`` /
  ``z_p, y_mask, g`` / ``z_in, g``) and output order, so a MoeVS-style consumer can swap the runtime graph by graph;
* ``StageSession`` — the consumer: same call signature and defaults as ``OnnxInferenceSession.__call__`` (numpy RNG seeded with
  ``seed``, ``zinput`` scaled by ``sdp_noise_scale``, ``z_p`` noise by ``seq_noise_scale``), with the length regulation
  between the stages done by this package's own kernels (``decode(...)`` without the flow/dec part is not exposed, so the
  glue here is the gather form of ``generate_path`` on the device).

Like everything in this package it has no CPU path: the runners raise without a GPU.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import hparams as H

STAGES = ("emb_g", "enc", "sdp", "dp", "flow", "dec")
INPUT_NAMES = {
    "emb_g": ("sid",),
    "enc": ("x", "t", "language", "bert_0", "bert_1", "bert_2", "g"),      # + optional "x_lengths" (exported, unused by the consumer)
    "sdp": ("x", "x_mask", "zin", "g"),
    "dp": ("x", "x_mask", "g"),
    "flow": ("z_p", "y_mask", "g"),
    "dec": ("z_in", "g"),
}
OUTPUT_NAMES = {"emb_g": ("g",), "enc": ("xout", "m_p", "logs_p", "x_mask"), "sdp": ("logw",), "dp": ("logw",), "flow": ("z",),
                "dec": ("o",)}


class StageRunner:
    """One exported graph of the reference on the HIP path; ``run`` has onnxruntime's calling convention."""

    def __init__(self, model, stage: str):
        if stage not in STAGES:
            raise ValueError(f"unknown stage {stage!r}; one of {STAGES}")
        self.model, self.stage = model, stage

    def get_inputs(self) -> Sequence[str]:
        return INPUT_NAMES[self.stage]

    def get_outputs(self) -> Sequence[str]:
        return OUTPUT_NAMES[self.stage]

    def _call(self, feeds: Dict[str, torch.Tensor]) -> List[torch.Tensor]:
        m, s = self.model, self.stage
        if s == "emb_g":
            return [m.stage_emb_g(feeds["sid"])]
        if s == "enc":
            x = feeds["x"]
            berts = []
            for k in ("bert_0", "bert_1", "bert_2"):
                b = feeds[k]
                if b.dim() == 2:                       # the exported graph takes [T, 1024] (batch 1, models_onnx.py:335-341)
                    b = b.transpose(0, 1).unsqueeze(0)
                berts.append(b)
            return list(m.stage_enc_p(x, feeds["t"], feeds["language"], berts[0], berts[1], berts[2], feeds["g"],
                                      x_lengths=feeds.get("x_lengths")))
        if s == "sdp":
            return [m.stage_sdp(feeds["x"], feeds["x_mask"], feeds["zin"], feeds["g"])]
        if s == "dp":
            return [m.stage_dp(feeds["x"], feeds["x_mask"], feeds["g"])]
        if s == "flow":
            return [m.stage_flow(feeds["z_p"], None, feeds["g"], y_mask=feeds["y_mask"])]
        return [m.stage_generator(feeds["z_in"], None, feeds["g"])]

    def run(self, output_names: Optional[Sequence[str]], feeds: Dict[str, np.ndarray]) -> List[np.ndarray]:
        missing = [k for k in INPUT_NAMES[self.stage] if k not in feeds]
        if missing:
            raise ValueError(f"stage {self.stage}: missing inputs {missing}")
        dev = self.model.device
        t = {k: (v if isinstance(v, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(v))).to(dev) for k, v in feeds.items()}
        outs = dict(zip(OUTPUT_NAMES[self.stage], self._call(t)))
        names = list(output_names) if output_names else list(OUTPUT_NAMES[self.stage])
        return [outs[n].detach().cpu().numpy() for n in names]


class StageSession:
    """The reference's ``OnnxInferenceSession`` (onnx_modules/V230_OnnxInference/__init__.py:36-126) over ``StageRunner``s."""

    def __init__(self, model):
        self.model = model
        for s in STAGES:
            setattr(self, s, StageRunner(model, s))

    def __call__(self, seq, tone, language, bert_zh, bert_jp, bert_en, sid, seed=114514, seq_noise_scale=0.8,
                 sdp_noise_scale=0.6, length_scale=1.0, sdp_ratio=0.0):
        seq, tone, language = (np.atleast_2d(np.asarray(a)).astype(np.int64) for a in (seq, tone, language))
        g = self.emb_g.run(None, {"sid": np.asarray(sid).astype(np.int64)})[0][..., None]
        x, m_p, logs_p, x_mask = self.enc.run(None, {
            "x": seq, "t": tone, "language": language, "bert_0": np.asarray(bert_zh, np.float32),
            "bert_1": np.asarray(bert_jp, np.float32), "bert_2": np.asarray(bert_en, np.float32), "g": g.astype(np.float32)})
        np.random.seed(seed)
        zin = (np.random.randn(x.shape[0], 2, x.shape[2]) * sdp_noise_scale).astype(np.float32)
        logw = self.sdp.run(None, {"x": x, "x_mask": x_mask, "zin": zin, "g": g})[0] * sdp_ratio \
            + self.dp.run(None, {"x": x, "x_mask": x_mask, "g": g})[0] * (1 - sdp_ratio)
        w_ceil = np.ceil(np.exp(logw) * x_mask * length_scale)
        y_lengths = np.clip(w_ceil.sum((1, 2)), 1.0, 100000).astype(np.int64)
        Ty = int(y_lengths.max())
        y_mask = (np.arange(Ty)[None, :] < y_lengths[:, None])[:, None, :].astype(np.float32)
        # length regulation, gather form: frame j belongs to the symbol whose cumulative duration first exceeds j
        cum = np.cumsum(w_ceil[:, 0], -1)
        idx = (np.arange(Ty)[None, :, None] >= cum[:, None, :]).sum(-1).clip(max=x.shape[2] - 1)
        m_e = np.take_along_axis(m_p, idx[:, None, :].repeat(m_p.shape[1], 1), 2) * y_mask
        logs_e = np.take_along_axis(logs_p, idx[:, None, :].repeat(m_p.shape[1], 1), 2) * y_mask
        z_p = m_e + np.random.randn(*m_e.shape) * np.exp(logs_e) * seq_noise_scale
        z = self.flow.run(None, {"z_p": z_p.astype(np.float32), "y_mask": y_mask, "g": g})[0]
        return self.dec.run(None, {"z_in": z.astype(np.float32), "g": g})[0]
