"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, and its host-only entry points
(create / load_tensor / pack_weights / workspace_bytes) behave — no compute calls (there is no GPU here)."""
import ctypes as C
import os
import re

import pytest
import torch

from bert_vits2_amd import hparams as H, lib as L, models, schema
from oracle import bv2_oracle as O
from tests.helpers import ROOT, cached_state_dict


def _declared(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bv2_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    names = _declared("bv2.h") + _declared("bv2_testing.h")
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"libbv2.so does not export {n}"
    assert sorted(s[0] for s in L.SYMBOLS) == _declared("bv2.h"), "lib.py binding list drifted from include/bv2.h"
    assert lib.bv2_abi_version() == 1


def test_create_rejects_bad_config_with_message():
    lib = L.load()
    cfg = L.make_config(H.default_v23())
    cfg.n_heads = 5
    h = C.c_void_p()
    assert lib.bv2_create(C.byref(cfg), C.byref(h)) != 0
    assert b"n_heads" in lib.bv2_last_error(None)
    cfg = L.make_config(H.default_v23())
    cfg.struct_bytes = 12
    assert lib.bv2_create(C.byref(cfg), C.byref(h)) != 0


def test_compute_calls_fail_loudly_without_weights_or_gpu():
    lib = L.load()
    cfg = L.make_config(H.default_v23())
    h = C.c_void_p()
    assert lib.bv2_create(C.byref(cfg), C.byref(h)) == 0
    ein, eout = L.EncodeIn(), L.EncodeOut()
    rc = lib.bv2_encode_durations(h, None, C.byref(ein), C.byref(eout), C.c_void_p(8), 8)
    assert rc != 0 and b"no weights attached" in lib.bv2_last_error(h)
    assert lib.bv2_workspace_bytes(h, 1, 128, 384) > 0
    assert lib.bv2_workspace_bytes(h, 4, 128, 384) > lib.bv2_workspace_bytes(h, 1, 128, 384)
    lib.bv2_destroy(h)
    m = models.from_hparams(H.default_v23())
    with pytest.raises(RuntimeError, match="GPU"):
        m.infer(torch.zeros(1, 4, dtype=torch.long), torch.tensor([4]), torch.tensor([0]), torch.zeros(1, 4, dtype=torch.long),
                torch.zeros(1, 4, dtype=torch.long), torch.zeros(1, 1024, 4), torch.zeros(1, 1024, 4), torch.zeros(1, 1024, 4))


@pytest.mark.parametrize("tf", [True, False])
def test_shim_state_dict_schema_and_tolerant_load(tf):
    hp = H.default_v23(use_transformer_flow=tf)
    m = models.from_hparams(hp)
    assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == dict(schema.param_shapes(hp))
    for attr in ("emb_g", "enc_p", "sdp", "dp", "flow", "dec"):      # attribute access used by the ONNX exporter
        assert hasattr(m, attr)
    sd = dict(cached_state_dict(hp, 0))
    sd["enc_q.pre.weight"] = torch.zeros(192, 1025, 1)              # training-only key, as in a real G_*.pth
    m.load_state_dict(sd, strict=False)
    assert torch.equal(m.state_dict()["dec.conv_post.weight"], sd["dec.conv_post.weight"])


def test_pack_weights_fold_equivalence_and_missing_keys():
    """bv2_pack_weights folds weight_norm exactly like torch (g*v/||v||, dim 0; C_in for ConvTranspose1d): packing the
    g/v form and packing the already-folded `.weight` form (Generator.remove_weight_norm) give the same blob."""
    hp = H.default_v23(use_transformer_flow=False)       # residual flow: WN layers are weight-normed too
    sd = cached_state_dict(hp, 0)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    blob1 = m.pack_host_blob()
    assert blob1.numel() == L.load().bv2_packed_bytes(m._handle) and blob1.numel() > 100e6
    blob1b = m.pack_host_blob()
    assert torch.equal(blob1, blob1b)                    # deterministic / idempotent

    lib = L.load()
    cfg = L.make_config(hp)
    h = C.c_void_p()
    assert lib.bv2_create(C.byref(cfg), C.byref(h)) == 0
    folded = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            p = k[: -len(".weight_g")]
            folded[p + ".weight"] = O.fold_weight_norm(sd, p)
        elif not k.endswith(".weight_v"):
            folded[k] = v
    for k, v in folded.items():
        t = v.contiguous()
        shp = (C.c_int64 * t.dim())(*t.shape)
        assert lib.bv2_load_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), shp, t.dim(), L.F32) in (0, 1)
    n = lib.bv2_packed_bytes(h)
    blob2 = torch.empty(n, dtype=torch.uint8)
    assert lib.bv2_pack_weights(h, C.c_void_p(blob2.data_ptr()), n) == 0, lib.bv2_last_error(h)
    a, b = blob1[256:].view(torch.float32), blob2[256:].view(torch.float32)
    assert a.abs().sum() > 0
    assert (a - b).abs().max().item() <= 2e-6 * a.abs().max().item()
    lib.bv2_destroy(h)

    # a missing tensor is reported by name
    h = C.c_void_p()
    assert lib.bv2_create(C.byref(cfg), C.byref(h)) == 0
    for k, v in sd.items():
        if k == "dec.ups.2.weight_g":
            continue
        t = v.contiguous()
        shp = (C.c_int64 * t.dim())(*t.shape)
        lib.bv2_load_tensor(h, k.encode(), C.c_void_p(t.data_ptr()), shp, t.dim(), L.F32)
    assert lib.bv2_pack_weights(h, C.c_void_p(blob2.data_ptr()), n) != 0
    assert b"dec.ups.2" in lib.bv2_last_error(h)
    lib.bv2_destroy(h)
