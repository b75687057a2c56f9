"""Shared helpers for the parity tests (test infrastructure)."""
import json
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    arrays = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    return meta, arrays


def rms(t):
    return float(t.double().pow(2).mean().sqrt())


def valid_wave_mask(y_lengths, hop, S):
    """Samples past y_lengths*hop are convolution bleed of the unmasked decoder."""
    idx = torch.arange(S)[None, :]
    return (idx < (y_lengths.long() * hop)[:, None])[:, None, :]


_SD_CACHE = {}


def cached_state_dict(hp, seed, **kw):
    from bert_vits2_amd import synth
    import dataclasses
    # every field that shapes the checkpoint (round 5: resblock type / kernel sizes, inter_channels, ... — not only the flow variant)
    key = (repr(dataclasses.astuple(hp)), seed, tuple(sorted(kw.items())))
    if key not in _SD_CACHE:
        _SD_CACHE[key] = synth.synthetic_state_dict(hp, seed, **kw)
    return _SD_CACHE[key]
