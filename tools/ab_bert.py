#!/usr/bin/env python3
"""Same-box A/B of bv2_bert_set_option switches on the BERT feature extractor (run ON THE GPU BOX):
    python tools/ab_bert.py "prefetch=0" "prefetch=3" ...      each argument = one run, in the given order (list the baseline first and last)
chinese-roberta-wwm-ext-large shape (24 x 1024, 22 layers run), one 53-token sentence, seeded synthetic weights; ms per sentence."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bert_vits2_amd import bert_synth as BS  # noqa: E402
from bert_vits2_amd.bert_encoder import BertEncoder  # noqa: E402


def main():
    cfg, S, layers = BS.LARGE, 53, 22
    sd = BS.bert_state_dict(cfg, 0, layers=layers)
    enc = BertEncoder(**cfg).load_state_dict(sd, device="cuda")
    ids, _ = BS.synthetic_inputs(cfg, [S], 0)
    ids = ids.cuda()
    ref = None
    for spec in sys.argv[1:] or [""]:
        for kv in filter(None, spec.split(",")):
            k, v = kv.split("=")
            enc.set_option(k, int(v))
        for _ in range(5):
            out = enc(ids)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(40):
                out = enc(ids)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 40 * 1e3)
        if ref is None:
            ref = out.clone()
        same = bool(torch.equal(out, ref))
        print(f"{spec or '(defaults)':24s} {sorted(ts)[1]:8.4f} ms per sentence  (min {min(ts):.4f} max {max(ts):.4f})  output identical to the first run: {same}", flush=True)


if __name__ == "__main__":
    main()
