"""CPU: the BERT oracle restatement (oracle/bert_oracle.py) against the goldens the REAL transformers.BertModel produced
(oracle/gen_bert_golden.py -> tests/golden/bert_*.npz), and the host side of the device extractor's C ABI (include/bv2_bert.h):
config validation, key routing / shape checks of bv2_bert_pack_tensor, completeness accounting.  No GPU compute."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import bert_oracle as BO
from oracle.gen_bert_golden import CASES

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_the_real_bertmodel(name):
    cfg, lengths, seed, use_tt = CASES[name]
    g = np.load(os.path.join(GOLD, f"bert_{name}.npz"))
    sd = BO.synthetic_state_dict(cfg, seed)
    assert hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).hexdigest() == str(g["weights_sha256"])
    ids, tt, ln = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["token_type_ids"]), torch.from_numpy(g["lengths"])
    n = cfg["num_hidden_layers"]
    got = BO.hidden_state(sd, cfg, ids, n - 2, token_type_ids=tt, lengths=ln)
    ref = torch.from_numpy(g["hidden_m3"])
    valid = (torch.arange(ids.shape[1])[None, :] < ln[:, None])[..., None]
    assert ((got - ref).abs() * valid).max().item() < 2e-5          # fp32 round-off between two formulations of the same math
    emb = BO.hidden_state(sd, cfg, ids, 0, token_type_ids=tt, lengths=ln)
    assert (emb - torch.from_numpy(g["hidden_0"])).abs().max().item() < 2e-6
    # fp64 restatement agrees too (the GPU tests use it as the tighter reference)
    got64 = BO.hidden_state(sd, cfg, ids, n - 2, token_type_ids=tt, lengths=ln, dtype=torch.float64)
    assert ((got64.float() - ref).abs() * valid).max().item() < 2e-5


def _lib():
    from bert_vits2_amd import lib as L
    return L, L.load()


def test_cabi_config_validation_and_pack_accounting():
    L, lib = _lib()
    mk = lambda **kw: L.BertConfig(C.sizeof(L.BertConfig), *[{**dict(v=97, h=128, nh=2, i=384, mp=48, tv=2, lr=3), **kw}[k]
                                                              for k in ("v", "h", "nh", "i", "mp", "tv", "lr")], 1e-12)
    h = C.c_void_p()
    for bad in (dict(h=100), dict(nh=3), dict(h=2048), dict(lr=0), dict(h=128, nh=8)):      # head_dim 16 is not a multiple of 32
        assert lib.bv2_bert_create(C.byref(mk(**bad)), C.byref(h)) != 0, bad
        assert b"bv2_bert_create" in lib.bv2_bert_last_error(None)
    cfg = mk()
    assert lib.bv2_bert_create(C.byref(cfg), C.byref(h)) == 0
    n = lib.bv2_bert_packed_bytes(h)
    assert n > 4 * (97 * 128 + 3 * (4 * 128 * 128 + 2 * 128 * 384))
    blob = torch.zeros(n // 4)
    sd = BO.synthetic_state_dict(BO.TINY, 0)                 # 5 layers; this handle runs 3
    codes = {}
    for k, v in sd.items():
        t = v.contiguous()
        shp = (C.c_int64 * t.dim())(*t.shape)
        codes[k] = lib.bv2_bert_pack_tensor(h, C.c_void_p(blob.data_ptr()), n, ("bert." + k).encode(), C.c_void_p(t.data_ptr()), shp, t.dim())
    assert all(c == (1 if k.startswith(("encoder.layer.3.", "encoder.layer.4.")) else 0) for k, c in codes.items()), codes
    assert lib.bv2_bert_missing(h) == 0
    # unknown keys are skipped, wrong shapes are errors
    t = torch.zeros(5, 7)
    shp = (C.c_int64 * 2)(5, 7)
    assert lib.bv2_bert_pack_tensor(h, C.c_void_p(blob.data_ptr()), n, b"cls.predictions.bias", C.c_void_p(t.data_ptr()), shp, 2) == 1
    assert lib.bv2_bert_pack_tensor(h, C.c_void_p(blob.data_ptr()), n, b"encoder.layer.0.output.dense.weight", C.c_void_p(t.data_ptr()), shp, 2) == -3
    assert b"shape mismatch" in lib.bv2_bert_last_error(h)
    # the query rows carry 1/sqrt(head_dim) (= 1/8 at head_dim 64: exact), the key rows are stored as they are
    assert blob.abs().sum().item() > 0
    # a second handle reports what is missing
    h2 = C.c_void_p()
    assert lib.bv2_bert_create(C.byref(cfg), C.byref(h2)) == 0
    assert lib.bv2_bert_missing(h2) == 5 + 3 * 16
    assert b"missing tensors" in lib.bv2_bert_last_error(h2)
    assert lib.bv2_bert_workspace_bytes(h, 2, 40) > 0
    lib.bv2_bert_destroy(h)
    lib.bv2_bert_destroy(h2)


def test_encoder_wrapper_refuses_deberta_and_cpu():
    from bert_vits2_amd.bert_encoder import BertEncoder
    with pytest.raises(NotImplementedError):
        BertEncoder(model_type="deberta-v2")
    enc = BertEncoder(**BO.TINY)
    assert enc.layers_run == 3
    with pytest.raises(RuntimeError):
        enc.load_state_dict(BO.synthetic_state_dict(BO.TINY, 0), device="cpu")
