"""Import the REAL reference (``/root/reference``) — build-container only, test infrastructure.

``/root/reference`` does not exist on the GPU box, so nothing that runs there may call this.
It is used by ``oracle/gen_golden.py`` (fixture generation) and by the CPU-only tests that
re-validate the restatement against the live reference when it happens to be present.

Two ``sys.modules`` stubs are needed: ``monotonic_align`` (numba is absent,
training only) and ``text`` (its ``__init__`` downloads BERT models at import).  The reference's source is
imported from where it lies; nothing is copied.
"""
from __future__ import annotations

import contextlib
import importlib.util
import json
import os
import sys
import types

import torch

REF = os.environ.get("BV2_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF, "models.py"))


_models = None


def reference_models():
    """Return the reference ``models`` module (unmodified source)."""
    global _models
    if _models is not None:
        return _models
    if not available():
        raise RuntimeError("reference not present at %s" % REF)
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.modules.setdefault("monotonic_align", types.ModuleType("monotonic_align"))
    spec = importlib.util.spec_from_file_location("text.symbols", os.path.join(REF, "text", "symbols.py"))
    sym = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sym)
    text = types.ModuleType("text")
    text.__path__ = []
    for k in dir(sym):
        if not k.startswith("__"):
            setattr(text, k, getattr(sym, k))
    sys.modules["text"] = text
    sys.modules["text.symbols"] = sym
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import models  # noqa: the reference's models.py
    _models = models
    return models


def reference_config() -> dict:
    with open(os.path.join(REF, "configs", "config.json"), "r", encoding="utf-8") as f:
        return json.load(f)


def build_reference_net(hp, state_dict):
    """Reference ``SynthesizerTrn`` (as reference infer.py:95-101 builds it) loaded with ``state_dict``."""
    import warnings
    models = reference_models()
    cfg = reference_config()
    model_kwargs = dict(cfg["model"])
    model_kwargs["use_transformer_flow"] = hp.use_transformer_flow
    model_kwargs["n_flow_layer"] = hp.n_flow_layer
    model_kwargs["n_layers_trans_flow"] = hp.n_layers_trans_flow
    model_kwargs["resblock"] = str(hp.resblock)                      # "1" = modules.ResBlock1, anything else ResBlock2 (models.py:508)
    model_kwargs["resblock_kernel_sizes"] = [int(k) for k in hp.resblock_kernel_sizes]
    model_kwargs["resblock_dilation_sizes"] = [[int(v) for v in d] for d in hp.resblock_dilation_sizes]
    # every width / depth the shim's HParams carries (a case may describe a narrower model than configs/config.json)
    for k in ("inter_channels", "hidden_channels", "filter_channels", "n_heads", "n_layers", "kernel_size", "upsample_initial_channel",
              "gin_channels"):
        model_kwargs[k] = int(getattr(hp, k))
    model_kwargs["upsample_rates"] = [int(v) for v in hp.upsample_rates]
    model_kwargs["upsample_kernel_sizes"] = [int(v) for v in hp.upsample_kernel_sizes]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        net = models.SynthesizerTrn(hp.n_vocab, hp.spec_channels, hp.segment_size,
                                    n_speakers=hp.n_speakers, **model_kwargs).eval()
    missing, unexpected = net.load_state_dict(state_dict, strict=False)
    bad = [k for k in missing if not (k.startswith("enc_q.") or k.startswith("sdp.post_"))]
    if bad or unexpected:
        raise RuntimeError(f"schema mismatch: missing={bad[:8]} unexpected={list(unexpected)[:8]}")
    return net


@contextlib.contextmanager
def injected_noise(noise_w: torch.Tensor, noise_z: torch.Tensor):
    """Hand the reference's two RNG draws pre-generated buffers: the first
    ``torch.randn`` call inside ``infer`` (models.py:248-251) gets ``noise_w``; the ``torch.randn_like`` at
    models.py:1071 gets ``noise_z[:, :, :T_y]``."""
    real_randn, real_like = torch.randn, torch.randn_like
    state = {"w": 0, "z": 0}

    def fake_randn(*size, **kw):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else tuple(size)
        assert shape == tuple(noise_w.shape), (shape, noise_w.shape)
        state["w"] += 1
        return noise_w.clone()

    def fake_like(t, **kw):
        assert t.shape[:2] == noise_z.shape[:2] and t.shape[2] <= noise_z.shape[2], (t.shape, noise_z.shape)
        state["z"] += 1
        return noise_z[:, :, : t.shape[2]].clone()

    torch.randn, torch.randn_like = fake_randn, fake_like
    try:
        yield state
    finally:
        torch.randn, torch.randn_like = real_randn, real_like


@torch.no_grad()
def reference_infer(net, batch, noise_w, noise_z, **kw):
    """Run the reference's own ``infer`` with injected noise; returns its tuple plus hook taps."""
    taps = {}

    def hook(name):
        def f(mod, inp, out):
            taps[name] = out
        return f

    hs = [net.enc_p.register_forward_hook(hook("enc_p")), net.sdp.register_forward_hook(hook("sdp")),
          net.dp.register_forward_hook(hook("dp"))]
    try:
        with injected_noise(noise_w, noise_z) as st:
            o, attn, y_mask, (z, z_p, m_p, logs_p) = net.infer(
                batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"],
                batch["bert"], batch["ja_bert"], batch["en_bert"], **kw)
        assert st["w"] == 1 and st["z"] == 1, st
    finally:
        for h in hs:
            h.remove()
    ex, em, el, xm = taps["enc_p"]
    sdp_ratio = kw.get("sdp_ratio", 0)
    logw = taps["sdp"] * sdp_ratio + taps["dp"] * (1 - sdp_ratio)
    w_ceil = torch.ceil(torch.exp(logw) * xm * kw.get("length_scale", 1))
    return dict(o=o, attn=attn, y_mask=y_mask, z=z, z_p=z_p, m_p=m_p, logs_p=logs_p, enc_x=ex, enc_m=em,
                enc_logs=el, x_mask=xm, logw=logw, logw_sdp=taps["sdp"], logw_dp=taps["dp"], w_ceil=w_ceil)


@torch.no_grad()
def reference_autocast_runs(net, ref, sid):
    """The reference's own reduced-precision behaviour, on the intermediate tensors of an fp32 run ``ref``
    (= reference_infer output): ``dec`` under ``torch.autocast("cpu", bfloat16)`` fed the fp32 ``z`` (BASELINE config 3's
    reference-side meaning), ``flow`` under ``torch.autocast("cpu", float16)`` fed the fp32 ``z_p`` (config 5), and the
    two decoders on that fp16-flow ``z``.  The reference source is not modified: autocast is entered around calls of its
    own sub-modules, exactly as a user would wrap them."""
    g = net.emb_g(sid).unsqueeze(-1)
    zin = ref["z"] * ref["y_mask"]
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o_bf16dec = net.dec(zin, g=g).float()
    with torch.autocast("cpu", dtype=torch.float16):
        z_f16 = net.flow(ref["z_p"], ref["y_mask"], g=g, reverse=True).float()
    o_f16flow = net.dec(z_f16 * ref["y_mask"], g=g)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        o_both = net.dec(z_f16 * ref["y_mask"], g=g).float()
    return dict(o_bf16dec=o_bf16dec, z_f16flow=z_f16, o_f16flow=o_f16flow, o_f16flow_bf16dec=o_both)


@torch.no_grad()
def reference_seeded_infer(net, batch, seed, **kw):
    """``torch.manual_seed(seed); net.infer(...)`` with NO injection: the reference draws its own noise from the global
    CPU generator.  The two draws are recorded on the way (the real ``torch.randn`` / ``torch.randn_like`` are called and
    their results kept), so a fixture can pin both the noise and the outputs of a seeded reference run (the RNG contract
    of SURVEY.md §8b)."""
    real_randn, real_like = torch.randn, torch.randn_like
    rec = {}

    def rec_randn(*a, **k):
        t = real_randn(*a, **k)
        rec.setdefault("noise_w", t.clone())
        return t

    def rec_like(t, **k):
        r = real_like(t, **k)
        rec.setdefault("noise_z", r.clone())
        rec.setdefault("noise_z_strides", tuple(t.stride()))
        return r

    torch.manual_seed(seed)
    torch.randn, torch.randn_like = rec_randn, rec_like
    try:
        o, attn, y_mask, (z, z_p, m_p, logs_p) = net.infer(
            batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"],
            batch["bert"], batch["ja_bert"], batch["en_bert"], **kw)
    finally:
        torch.randn, torch.randn_like = real_randn, real_like
    return dict(o=o, attn=attn, y_mask=y_mask, z=z, z_p=z_p, noise_w=rec["noise_w"], noise_z=rec["noise_z"].contiguous(),
                noise_z_strides=torch.tensor(rec["noise_z_strides"]))
