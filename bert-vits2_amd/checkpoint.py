"""Checkpoint ingest for the drop-in ``SynthesizerTrn``.

``load_checkpoint`` has the signature, return value and tolerance rules of reference ``utils.load_checkpoint``
(utils.py:65-120), so ``infer.get_net_g`` (infer.py:84-104) can call either; on top of it this loader accepts the
checkpoint variants that exist in the wild for this model family:

* fp16 "release" checkpoints written by ``compress_model.py:23-63`` (``enc_q.*`` dropped, tensors ``.half()``) —
  widened to fp32 (the packer keeps fp32 master weights and derives its own bf16 / fp16 streams);
* checkpoints saved after ``Generator.remove_weight_norm`` (models.py:559-564) or the new parametrization API: a folded
  ``<conv>.weight`` (or ``parametrizations.weight.original0/1``) instead of ``weight_g`` / ``weight_v`` — re-expressed as
  ``weight_v = w``, ``weight_g = ||w||`` (norm over all dims but 0), which folds back to exactly ``w``;
* DDP checkpoints with a ``module.`` prefix;
* old checkpoints without ``ja_bert_proj`` / ``en_bert_proj``: zero-filled with a warning, as the reference does
  (utils.py:103-108); any other missing key is reported and keeps the model's current value.

``save_packed`` / ``load_packed`` cache the packed device blob (weight-norm folded, MFMA fragment order, bf16 / fp16
streams) on disk: start-up then skips the fold + repack (the blob header carries the config hash, so a blob packed for
another model is rejected by ``bv2_attach_weights``).
"""
from __future__ import annotations

import logging
import os
from typing import Dict, Optional, Tuple

import torch

logger = logging.getLogger(__name__)

_PARAM_G = ".parametrizations.weight.original0"
_PARAM_V = ".parametrizations.weight.original1"


def _normalise_keys(saved: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    out = {}
    for k, v in saved.items():
        if k.startswith("module."):
            k = k[len("module."):]
        if k.endswith(_PARAM_G):
            k = k[:-len(_PARAM_G)] + ".weight_g"
        elif k.endswith(_PARAM_V):
            k = k[:-len(_PARAM_V)] + ".weight_v"
        out[k] = v
    return out


def _norm_dim0(w: torch.Tensor) -> torch.Tensor:
    return w.reshape(w.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (w.dim() - 1)))


def adapt_state_dict(saved: Dict[str, torch.Tensor], want: Dict[str, torch.Tensor]) -> Tuple[Dict[str, torch.Tensor], list]:
    """Map a checkpoint's ``model`` dict onto the shim's parameter names/shapes (fp32).  Returns (state_dict, missing)."""
    saved = _normalise_keys(saved)
    new, missing = {}, []
    for k, cur in want.items():
        t = saved.get(k)
        if t is None and k.endswith(".weight_v") and (k[:-2] in saved):          # folded conv: <p>.weight
            t = saved[k[:-2]]
        if t is None and k.endswith(".weight_g") and (k[:-2] in saved):
            t = _norm_dim0(saved[k[:-2]].float())
        if t is not None and tuple(t.shape) == tuple(cur.shape):
            new[k] = t.detach().to(torch.float32)
            continue
        if t is not None:
            logger.error("%s has shape %s in the checkpoint, the model needs %s", k, tuple(t.shape), tuple(cur.shape))
        # reference utils.py:101-110
        if "ja_bert_proj" in k or "en_bert_proj" in k:
            logger.warning("Seems you are using the old version of the model, the %s is automatically set to zero for "
                           "backward compatibility", k)
            new[k] = torch.zeros_like(cur, dtype=torch.float32)
        else:
            logger.error("%s is not in the checkpoint", k)
            new[k] = cur.detach().to(torch.float32)
            missing.append(k)
    return new, missing


def load_checkpoint(checkpoint_path, model, optimizer=None, skip_optimizer=False, *, trust_pickle: bool = False):
    """reference utils.py:65-120: returns ``(model, optimizer, learning_rate, iteration)``.

    Checkpoints of this model family hold tensors, dicts and scalars only (utils.py:123-139), so the file is read with
    ``weights_only=True`` — a third-party ``.pth`` cannot run code at load time.  ``trust_pickle=True`` (or the environment
    variable ``BV2_TRUST_PICKLE=1``) opts in to the unrestricted unpickler for legacy files that embed other objects."""
    assert os.path.isfile(checkpoint_path)
    if trust_pickle or os.environ.get("BV2_TRUST_PICKLE") == "1":
        logger.warning("loading %s with the unrestricted unpickler (trust_pickle): only do this for files you trust", checkpoint_path)
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    else:
        ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    iteration = ckpt.get("iteration", 0)
    learning_rate = ckpt.get("learning_rate", 0.0)
    if optimizer is not None and not skip_optimizer and ckpt.get("optimizer") is not None:
        optimizer.load_state_dict(ckpt["optimizer"])
    target = model.module if hasattr(model, "module") else model
    want = {k: v for k, v in target.state_dict().items()}
    new, missing = adapt_state_dict(ckpt["model"], want)
    target.load_state_dict(new, strict=False)
    target.last_missing_keys = missing
    logger.info("Loaded checkpoint '%s' (iteration %s)", checkpoint_path, iteration)
    return model, optimizer, learning_rate, iteration


# --------------------------------------------------------------------------------------------------------------------
def save_packed(model, path: str) -> int:
    """Write the packed weight blob of ``model`` (host pack; works without a GPU).  Returns the number of bytes."""
    blob = model.pack_host_blob()
    tmp = path + ".tmp"
    with open(tmp, "wb") as f:
        f.write(blob.numpy().tobytes())
    os.replace(tmp, path)
    return blob.numel()


def load_packed(model, path: str, device: Optional[torch.device] = None) -> None:
    """Attach a blob written by ``save_packed`` (must have been packed for the same hyper-parameters: the header's
    config hash is checked by the library).  Needs the GPU the model will run on."""
    import numpy as np
    dev = torch.device(device) if device is not None else model.device
    if dev.type != "cuda":
        raise RuntimeError("load_packed needs a GPU device (there is no CPU fallback)")
    lib = model._ensure_handle()
    want = int(lib.bv2_packed_bytes(model._handle))
    host = torch.from_numpy(np.fromfile(path, dtype=np.uint8))
    if host.numel() != want:
        raise ValueError(f"{path}: {host.numel()} bytes, this model's packed blob has {want}")
    with torch.cuda.device(dev):
        model.attach_blob(host.to(dev))
