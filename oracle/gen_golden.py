"""Generate ``tests/golden/*.npz`` by running the REAL reference — build-container only.

    python -m oracle.gen_golden            # from the repo root, needs /root/reference

For every case in ``oracle/cases.py`` this imports the reference's own ``models.py`` unmodified
(``oracle/ref_import.py``), loads the seeded synthetic checkpoint through ``load_state_dict``, injects the seeded
noise into the two RNG draws of ``infer`` (models.py:248-251, :1071) and stores what the reference returned.
The reference has no golden vectors of its own; these fixtures are what pins the oracle.
Determinism: torch CPU fp32, 1 thread, ``use_deterministic_algorithms``.
"""
from __future__ import annotations

import dataclasses
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bert_vits2_amd import schema, synth  # noqa: E402
from oracle import cases, ref_import  # noqa: E402


def net_key(hp, seed):
    """Cache key of a built reference net: EVERY hyper-parameter field plus the weight seed (round 5 keyed on the flow variant only, so a
    multi-case invocation handed `rb2_b2_t14` the ResBlock1 net and `narrow_b2_t18` the full-width one)."""
    return (repr(dataclasses.astuple(hp)), seed)


def generate_case(name, nets):
    """Run the real reference on one case; ``nets`` caches built reference nets across cases of one process.  Returns
    (arrays, meta, seeded-run arrays or None)."""
    hp, seed, batch, noise_w, noise_z, kw = cases.build_case(name)
    key = net_key(hp, seed)
    if key not in nets:
        sd = synth.synthetic_state_dict(hp, seed)
        nets[key] = (sd, ref_import.build_reference_net(hp, sd))
    sd, net = nets[key]
    ref = ref_import.reference_infer(net, batch, noise_w, noise_z, **kw)
    arrays = {k: ref[k].detach().float().numpy() for k in cases.GOLDEN_KEYS}
    arrays["y_lengths"] = ref["y_mask"].sum([1, 2]).long().numpy()
    if name in cases.AUTOCAST_CASES:
        ac = ref_import.reference_autocast_runs(net, ref, batch["sid"])
        arrays.update({k: ac[k].detach().float().numpy() for k in cases.AUTOCAST_KEYS})
    seeded = None
    if name == cases.SEEDED_CASE:
        sr = ref_import.reference_seeded_infer(net, batch, cases.SEEDED_SEED, **kw)
        seeded = {k: v.detach().numpy() for k, v in sr.items()}
    meta = dict(case=name, torch=torch.__version__, checksums=cases.weight_checksums(sd),
                o_rms=float(ref["o"].pow(2).mean().sqrt()), T_y=int(ref["y_mask"].shape[2]))
    return arrays, meta, seeded


def deterministic():
    torch.set_num_threads(1)
    torch.use_deterministic_algorithms(True)


def main():
    only = set(sys.argv[1:])                     # optional: regenerate only the named cases
    deterministic()
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    nets = {}
    for name in cases.CASES:
        if only and name not in only:
            continue
        arrays, meta, sa = generate_case(name, nets)
        if sa is not None:
            np.savez_compressed(os.path.join(out_dir, "seeded_" + name + ".npz"),
                                meta=json.dumps(dict(case=name, seed=cases.SEEDED_SEED, torch=torch.__version__)), **sa)
            print("seeded", name, {k: v.shape for k, v in sa.items()})
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=json.dumps(meta), **arrays)
        print(name, meta)
    if only:
        return
    # the reference's own state_dict schema (inference sub-networks), to pin bert_vits2_amd/schema.py
    hp_t, _, *_ = cases.build_case("zh_b1_t24")
    hp_w, _, *_ = cases.build_case("wn_b1_t16")
    sch = {}
    for tag, hp in (("transformer_flow", hp_t), ("residual_flow", hp_w)):
        net = nets[net_key(hp, 0)][1]
        sch[tag] = {k: list(v.shape) for k, v in net.state_dict().items()
                    if not (k.startswith("enc_q.") or k.startswith("sdp.post_"))}
    with open(os.path.join(out_dir, "reference_state_dict_schema.json"), "w") as f:
        json.dump(sch, f, indent=0, sort_keys=True)
    print("schema keys:", {k: len(v) for k, v in sch.items()})


if __name__ == "__main__":
    main()
