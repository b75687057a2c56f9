#!/usr/bin/env python3
"""Which hyper-parameter breaks the flow?  stage_flow (fp32) against the oracle's flow_reverse for one-at-a-time deviations from the released
config (run ON THE GPU BOX; test infrastructure: imports oracle/)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bert_vits2_amd import hparams as H, models, synth  # noqa: E402
from oracle import bv2_oracle as O  # noqa: E402

VARIANTS = {
    "released": {},
    "n_flow_layer=3": dict(n_flow_layer=3),
    "n_flow_layer=2": dict(n_flow_layer=2),
    "n_layers_trans_flow=3": dict(n_layers_trans_flow=3),
    "hidden=128,filter=512": dict(hidden_channels=128, filter_channels=512),
    "inter=128": dict(inter_channels=128),
    "gin=256": dict(gin_channels=256),
    "narrow": dict(hidden_channels=128, filter_channels=512, inter_channels=128, n_layers=4, n_layers_trans_flow=3, n_flow_layer=3, gin_channels=256),
}


def main():
    only = sys.argv[1:]
    for name, ov in VARIANTS.items():
        if only and name not in only:
            continue
        hp = H.default_v23(**ov)
        sd = synth.synthetic_state_dict(hp, seed=5)
        m = models.from_hparams(hp)
        m.load_state_dict(sd, strict=False)
        m = m.to("cuda").eval()
        B, Ty = 2, 45
        gen = torch.Generator().manual_seed(3)
        yl = torch.tensor([45, 29], dtype=torch.int64)
        ym = (torch.arange(Ty)[None, :] < yl[:, None])[:, None, :].float()
        z_p = torch.randn(B, hp.inter_channels, Ty, generator=gen) * ym
        g = torch.randn(B, hp.gin_channels, 1, generator=gen)
        ref = O.flow_reverse(sd, hp, z_p, ym, g)
        out = {}
        for fb in (1, 0):
            m.set_option("fused_boundary", fb)
            z = m.stage_flow(z_p.cuda(), yl.cuda(), g.cuda()).cpu()
            out[fb] = float(((z - ref) * ym).pow(2).mean().sqrt() / (ref * ym).pow(2).mean().sqrt())
        print(f"{name:26s} rel rms error: fused_boundary=1 {out[1]:.3e}   =0 {out[0]:.3e}", flush=True)
        del m


if __name__ == "__main__":
    main()
