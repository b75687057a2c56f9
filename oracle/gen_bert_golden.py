#!/usr/bin/env python3
"""TEST INFRASTRUCTURE — writes tests/golden/bert_*.npz with the REAL ``transformers`` model (run in the build container, where
``transformers`` is installed; the GPU box never needs it).  For each case: seeded synthetic weights (oracle/bert_oracle.py) are
loaded into ``AutoModelForMaskedLM.from_config(BertConfig(**cfg))`` — the class the reference instantiates — and the model runs exactly
as the reference calls it
(text/chinese_bert.py:34-37: tokenizer output -> ``model(**inputs, output_hidden_states=True)`` -> ``hidden_states[-3]``) and the
hidden state is stored together with the inputs and a checksum of the weights.

    python oracle/gen_bert_golden.py
"""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import bert_oracle as BO  # noqa: E402

CASES = {  # name: (config, lengths, weight seed, token types)
    "tiny_b1_s19": (BO.TINY, [19], 0, False),
    "tiny_b3_ragged": (BO.TINY, [33, 7, 40], 1, True),
    "mid_b2_s70": (BO.MID, [70, 52], 2, False),
}


DEBERTA_CASES = {  # name: (config, lengths, weight seed, which reference class)
    "v3_tiny_b1_s23": ("TINY_V3", [23], 0, "DebertaV2Model"),                 # english_bert_mock.py: DebertaV2Model
    "ja_tiny_b3_ragged": ("TINY_JA", [40, 9, 31], 1, "AutoModelForMaskedLM"),   # japanese_bert.py: AutoModelForMaskedLM, conv layer
    "v3_mid_b2_s150": ("MID_V3", [150, 97], 2, "DebertaV2Model"),             # > bucket/2 apart: the log-spaced buckets
}


def deberta_main():
    from transformers import AutoModelForMaskedLM, DebertaV2Config, DebertaV2Model
    from oracle import deberta_oracle as DO
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (cfg_name, lengths, seed, cls) in DEBERTA_CASES.items():
        cfg = getattr(DO, cfg_name)
        sd = DO.synthetic_state_dict(cfg, seed)
        if cls == "DebertaV2Model":
            m = DebertaV2Model(DebertaV2Config(**cfg)).eval()
            missing, unexpected = m.load_state_dict(sd, strict=False)
        else:
            m = AutoModelForMaskedLM.from_config(DebertaV2Config(**cfg)).eval()
            assert type(m).__name__ == "DebertaV2ForMaskedLM"
            missing, unexpected = m.load_state_dict({"deberta." + k: v for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.startswith(("cls.", "lm_predictions.")) or "position_ids" in k for k in missing), (missing, unexpected)
        g = torch.Generator().manual_seed(88 + seed)
        S = max(lengths)
        ids = torch.randint(1, cfg["vocab_size"], (len(lengths), S), generator=g)
        ln = torch.tensor(lengths, dtype=torch.int64)
        am = (torch.arange(S)[None, :] < ln[:, None]).long()
        ids = ids * am                                           # [PAD] = 0 beyond the sentence
        with torch.no_grad():
            res = m(input_ids=ids, attention_mask=am, output_hidden_states=True)
        hs = res["hidden_states"]
        assert len(hs) == cfg["num_hidden_layers"] + 1
        digest = hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).hexdigest()
        np.savez_compressed(os.path.join(out_dir, f"deberta_{name}.npz"), input_ids=ids.numpy(), lengths=ln.numpy(),
                            hidden_m3=hs[-3].numpy().astype(np.float32), hidden_0=hs[0].numpy().astype(np.float32),
                            hidden_1=hs[1].numpy().astype(np.float32), weights_sha256=np.array(digest), seed=np.array(seed))
        print("deberta", name, tuple(hs[-3].shape), float(hs[-3].abs().mean()), digest[:12])


def main():
    from transformers import AutoModelForMaskedLM, BertConfig
    out_dir = os.path.join(ROOT, "tests", "golden")
    for name, (cfg, lengths, seed, use_tt) in CASES.items():
        sd = BO.synthetic_state_dict(cfg, seed)
        # the class the reference instantiates (text/chinese_bert.py:31): AutoModelForMaskedLM -> BertForMaskedLM; the encoder's
        # tensors carry the "bert." prefix there, the MLM head (cls.*) keeps its random init — it never touches hidden_states
        m = AutoModelForMaskedLM.from_config(BertConfig(**cfg)).eval()
        assert type(m).__name__ == "BertForMaskedLM"
        missing, unexpected = m.load_state_dict({"bert." + k: v for k, v in sd.items()}, strict=False)
        assert not unexpected and all(k.startswith("cls.") or "position_ids" in k or "token_type_ids" in k for k in missing), (missing, unexpected)
        ids, ln = BO.synthetic_inputs(cfg, lengths, seed)
        S = ids.shape[1]
        am = (torch.arange(S)[None, :] < ln[:, None]).long()
        tt = ((torch.arange(S)[None, :] >= (ln[:, None] // 2)) & (am > 0)).long() if use_tt else torch.zeros_like(ids)
        with torch.no_grad():
            res = m(input_ids=ids, token_type_ids=tt, attention_mask=am, output_hidden_states=True)
        hs = res["hidden_states"]
        assert len(hs) == cfg["num_hidden_layers"] + 1
        h3 = hs[-3]
        digest = hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).hexdigest()
        np.savez_compressed(os.path.join(out_dir, f"bert_{name}.npz"), input_ids=ids.numpy(), token_type_ids=tt.numpy(),
                            lengths=ln.numpy(), hidden_m3=h3.numpy().astype(np.float32), hidden_0=hs[0].numpy().astype(np.float32),
                            weights_sha256=np.array(digest), seed=np.array(seed), cfg_hidden=np.array(cfg["hidden_size"]))
        print(name, tuple(h3.shape), float(h3.abs().mean()), digest[:12])


if __name__ == "__main__":
    main()
    deberta_main()
