"""CPU: the BERT oracle restatement (oracle/bert_oracle.py) against the goldens the REAL transformers.BertModel produced
(oracle/gen_bert_golden.py -> tests/golden/bert_*.npz), and the host side of the device extractor's C ABI (include/bv2_bert.h):
config validation, key routing / shape checks of bv2_bert_pack_tensor, completeness accounting.  No GPU compute."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import bert_oracle as BO, deberta_oracle as DO
from oracle.gen_bert_golden import CASES, DEBERTA_CASES

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


def test_encoder_wrapper_refuses_unsupported_variants_and_cpu():
    from bert_vits2_amd.bert_encoder import BertEncoder, _relative_index_table
    with pytest.raises(NotImplementedError):
        BertEncoder(model_type="roberta")
    with pytest.raises(NotImplementedError):
        BertEncoder(model_type="deberta-v2")                      # BERT's defaults are not a DeBERTa config
    with pytest.raises(NotImplementedError):
        BertEncoder(**dict(DO.TINY_V3, share_att_key=False), model_type="deberta-v2")
    with pytest.raises(NotImplementedError):
        BertEncoder(**dict(DO.TINY_JA, conv_act="tanh"), model_type="deberta-v2")
    d = BertEncoder(**DO.TINY_JA, model_type="deberta-v2")
    assert d.layers_run == 3 and d.arch == 1
    # the product's own table builder (no oracle import there) == the oracle's
    for cfg in (DO.TINY_V3, DO.MID_V3, DO.LARGE_V3):
        P = cfg["max_position_embeddings"]
        assert torch.equal(_relative_index_table(cfg["position_buckets"], P, DO.att_span(cfg), P), DO.relative_index_table(cfg, P).float())
    # packing a DeBERTa state_dict: derived tensors are added, rel_embeddings / encoder.LayerNorm are consumed by the wrapper
    _, lib = _lib()
    sd = d._with_deberta_derived({"deberta." + k: v for k, v in DO.synthetic_state_dict(DO.TINY_JA, 1).items()})
    n = lib.bv2_bert_packed_bytes(d._h)
    blob = torch.zeros(n // 4)
    codes = {}
    for k, v in sd.items():
        t = v.float().contiguous()
        shp = (C.c_int64 * max(t.dim(), 1))(*t.shape)
        codes[k] = lib.bv2_bert_pack_tensor(d._h, C.c_void_p(blob.data_ptr()), n, k.encode(), C.c_void_p(t.data_ptr()), shp, t.dim())
    unused = {k for k, c in codes.items() if c == 1}
    assert min(codes.values()) >= 0 and lib.bv2_bert_missing(d._h) == 0
    assert {"encoder.rel_embeddings.weight", "encoder.LayerNorm.weight", "encoder.LayerNorm.bias"} <= unused
    assert all(k.startswith(("encoder.layer.3.", "encoder.layer.4.", "encoder.rel_embeddings", "encoder.LayerNorm")) for k in unused), unused
    enc = BertEncoder(**BO.TINY)
    assert enc.layers_run == 3
    with pytest.raises(RuntimeError):
        enc.load_state_dict(BO.synthetic_state_dict(BO.TINY, 0), device="cpu")


@pytest.mark.parametrize("name", sorted(DEBERTA_CASES))
def test_deberta_oracle_matches_the_real_transformers_model(name):
    """oracle/deberta_oracle.py against goldens of DebertaV2Model / AutoModelForMaskedLM (the two classes the reference
    instantiates): embeddings, the output of layer 1 (which includes the Japanese model's ConvLayer) and hidden_states[-3]."""
    cfg_name, lengths, seed, _cls = DEBERTA_CASES[name]
    cfg = getattr(DO, cfg_name)
    g = np.load(os.path.join(GOLD, f"deberta_{name}.npz"))
    sd = DO.synthetic_state_dict(cfg, seed)
    assert hashlib.sha256(b"".join(sd[k].numpy().tobytes() for k in sorted(sd))).hexdigest() == str(g["weights_sha256"])
    ids, ln = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["lengths"])
    valid = (torch.arange(ids.shape[1])[None, :] < ln[:, None])[..., None]
    n = cfg["num_hidden_layers"]
    for layers, key, tol in ((0, "hidden_0", 2e-6), (1, "hidden_1", 2e-5), (n - 2, "hidden_m3", 5e-5)):
        got = DO.hidden_state(sd, cfg, ids, layers, lengths=ln)
        err = ((got - torch.from_numpy(g[key])).abs() * valid).max().item()
        assert err < tol, (key, err)
    got64 = DO.hidden_state(sd, cfg, ids, n - 2, lengths=ln, dtype=torch.float64).float()
    assert ((got64 - torch.from_numpy(g["hidden_m3"])).abs() * valid).max().item() < 5e-5


def test_relative_index_table_matches_the_pairwise_formulas():
    """The ONE index table the device kernel gathers both relative terms with == HF's c2p index and (transposed) p2c index."""
    for cfg in (DO.TINY_V3, DO.MID_V3, DO.LARGE_V3):
        S = min(cfg["max_position_embeddings"], 300)
        span = DO.att_span(cfg)
        tab = DO.relative_index_table(cfg, S)
        ids = torch.arange(S)
        rel = DO.log_bucket_position(ids[:, None] - ids[None, :], cfg["position_buckets"], cfg["max_position_embeddings"]).long()
        c2p = torch.clamp(rel + span, 0, 2 * span - 1)
        p2c_t = torch.clamp(-rel + span, 0, 2 * span - 1).T            # p2c is gathered on [k, q] and transposed
        pair = tab[(ids[:, None] - ids[None, :]) + S - 1]
        assert torch.equal(pair, c2p) and torch.equal(pair, p2c_t)
        d = tab[1:] - tab[:-1]
        assert bool(((d == 0) | (d == 1)).all())                        # monotone with slope <= 1: a 32x32 tile pair needs <= 63 rows


@pytest.mark.skipif(not os.path.isdir("/root/reference/bert"), reason="the reference tree (build container only)")
def test_configs_are_the_ones_the_reference_ships():
    import json
    ref = lambda d: json.load(open(f"/root/reference/bert/{d}/config.json"))
    z = ref("chinese-roberta-wwm-ext-large")
    assert all(z[k] == v for k, v in BO.LARGE.items()) and z["model_type"] == "bert" and z["hidden_act"] == "gelu"
    for name, mine in (("deberta-v3-large", DO.LARGE_V3), ("deberta-v2-large-japanese-char-wwm", DO.LARGE_JA)):
        c = ref(name)
        norm = lambda v: sorted(v.split("|")) if isinstance(v, str) and "|" in v else (sorted(v) if isinstance(v, list) else v)
        assert all(norm(c[k]) == norm(v) for k, v in mine.items()), name
        assert c["model_type"] == "deberta-v2" and c["hidden_act"] == "gelu"
