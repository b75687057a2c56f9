"""CPU: the C-ABI library loads, exports every symbol include/*.h declares, and its host-only entry points
(create / load_tensor / pack_weights / workspace_bytes) behave — no compute calls (there is no GPU here)."""
import math
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
    names = _declared("bv2.h") + _declared("bv2_testing.h") + _declared("bv2_bert.h")
    assert len(names) >= 30 and len(_declared("bv2_bert.h")) == 10
    for n in names:
        assert hasattr(lib, n), f"libbv2.so does not export {n}"
    assert sorted(s[0] for s in L.SYMBOLS) == sorted(_declared("bv2.h") + _declared("bv2_bert.h")), \
        "lib.py binding list drifted from include/bv2.h + include/bv2_bert.h"
    assert lib.bv2_abi_version() == 3


def test_library_exports_nothing_but_the_c_abi():
    """A drop-in C-ABI library exports `bv2_*` and nothing else: the linker version script csrc/libbv2.map keeps every C++ symbol of
    the executor / launchers (namespace bv2) local."""
    import subprocess
    lib = L.load()
    out = subprocess.run(["nm", "-D", "--defined-only", lib._name], check=True, capture_output=True, text=True).stdout
    exported = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert len(exported) >= 50
    stray = [n for n in exported if not n.startswith("bv2_")]
    assert not stray, stray[:10]
    declared = set(_declared("bv2.h") + _declared("bv2_testing.h") + _declared("bv2_bert.h"))
    undeclared = [n for n in exported if n not in declared]
    assert not undeclared, f"exported but declared in no header: {undeclared}"


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
    # the x6 weight planes (three bf16 planes whose SUM is the fp32 weight, kernels/conv_x6.hip) are compared as what they encode:
    # a fold difference of one fp32 ulp changes the low planes completely, but not their sum
    lib.bv2_test_x6_regions.restype = C.c_int
    lib.bv2_test_x6_regions.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    offs, cnts = (C.c_int64 * 256)(), (C.c_int64 * 256)()
    nreg = lib.bv2_test_x6_regions(h, offs, cnts, 256)
    assert 0 < nreg <= 256
    f1, f2 = blob1.view(torch.float32).clone(), blob2.view(torch.float32).clone()
    for i in range(nreg):
        sl = slice(offs[i], offs[i] + cnts[i])

        def decode(f):
            planes = (f[sl].view(torch.int16).to(torch.int32) << 16).view(torch.float32).view(-1, 3, 512).double()
            return planes.sum(1)
        w1, w2 = decode(f1), decode(f2)
        assert w1.abs().sum() > 0 and (w1 - w2).abs().max().item() <= 2e-6 * w1.abs().max().item()
        f1[sl] = 0
        f2[sl] = 0
    # the x3 regions (two fp16 planes of w * S_w behind 1 / S_w; kernels/conv_x6.hip NP = 2) encode the SAME weights as the x6 regions of the
    # same convs: (g0 + g1) / S_w == h1 + h2 + h3 up to 2^-23 of the tensor's largest weight (fp16 halves of a value below 2^15 leave <= 2^-24 of
    # it, or 2^-25 absolute in scaled units), S_w a power of two that puts that largest weight in [2^14, 2^15)
    lib.bv2_test_x3_regions.restype = C.c_int
    lib.bv2_test_x3_regions.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int]
    o3, c3 = (C.c_int64 * 256)(), (C.c_int64 * 256)()
    assert lib.bv2_test_x3_regions(h, o3, c3, 256) == nreg
    g1 = blob1.view(torch.float32)
    for i in range(nreg):
        w6 = (g1[offs[i]:offs[i] + cnts[i]].view(torch.int16).to(torch.int32) << 16).view(torch.float32).view(-1, 3, 512).double().sum(1)
        reg = g1[o3[i]:o3[i] + c3[i]]
        inv = float(reg[0])
        assert inv > 0 and math.log2(inv) == round(math.log2(inv))
        w3 = reg[64:].view(torch.float16).view(-1, 2, 512).double().sum(1)
        assert w3.shape == w6.shape
        wmax = w6.abs().max().item()
        assert 2.0 ** 14 <= wmax / inv < 2.0 ** 15
        assert (w3 * inv - w6).abs().max().item() <= 2.0 ** -23 * wmax
        f1[o3[i]:o3[i] + c3[i]] = 0
        f2[o3[i]:o3[i] + c3[i]] = 0
    a, b = f1[64:], f2[64:]
    assert a.abs().sum() > 0
    # fp32 regions agree to fold round-off; in the bf16 regions of the blob (two bf16 per 32-bit word) a fold difference
    # of 1e-7 can flip a bf16 rounding, which shows as <= 1 bf16 ulp (2^-7 relative) on a rare word
    diff = (a - b).abs()
    loose = diff > 2e-6 * a.abs().max().item()
    assert loose.float().mean().item() < 1e-3
    assert bool((diff[loose] <= 2.0 ** -7 * a[loose].abs() + 1e-30).all())
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


def _dump_cl(lib, h, blob, kind, i=0, j=0, d=0, e=0):
    dims = (C.c_int32 * 4)()
    args = (h, C.c_void_p(blob.data_ptr()), kind, i, j, d, e, dims)
    assert lib.bv2_test_dump_cl_conv(*args, None, None) == 0
    cin, cout, k, pad_left = list(dims)
    w = torch.empty(cout, cin, k)
    b = torch.empty(cout)
    assert lib.bv2_test_dump_cl_conv(*args, C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr())) == 0
    return w, b, pad_left


def test_bf16_generator_pack_matches_reference_convs():
    """The bf16 weight streams of the packed blob (bv2_model.cpp, cl_w_index) decode to bf16(folded reference weight), and
    the channels-last single-conv form of every ConvTranspose1d (C_out' = u*C_out over the union tap window) reproduces
    F.conv_transpose1d (reference models.py:510-522, 545) exactly when run as an ordinary conv and re-read as [B,C,L*u]."""
    import torch.nn.functional as F
    hp = H.default_v23()
    sd = cached_state_dict(hp, 0)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    blob = m.pack_host_blob()
    lib = L.load()
    lib.bv2_test_dump_cl_conv.restype = C.c_int
    h = m._handle
    bf = lambda t: t.to(torch.bfloat16).float()
    w, b, pl = _dump_cl(lib, h, blob, 0)
    assert pl == 3 and torch.equal(w, bf(sd["dec.conv_pre.weight"])) and torch.equal(b, sd["dec.conv_pre.bias"])
    nk = len(hp.resblock_kernel_sizes)
    for (i, j, d, e) in [(0, 0, 0, 0), (1, 2, 2, 0), (4, 1, 1, 1)]:
        w, b, pl = _dump_cl(lib, h, blob, 2, i, j, d, e)
        p = f"dec.resblocks.{i * nk + j}.convs{e + 1}.{d}"
        ref = bf(O.fold_weight_norm(sd, p))
        # the C++ fold (double sqrt) and torch's fp32 fold differ by <= 1 fp32 ulp, which may flip a rare bf16 rounding
        bad = w != ref
        assert bad.float().mean().item() < 1e-3 and bool(((w - ref).abs() <= 2.0 ** -7 * ref.abs()).all()), p
        assert torch.equal(b, sd[p + ".bias"])
        k, dil = hp.resblock_kernel_sizes[j], (hp.resblock_dilation_sizes[j][d] if e == 0 else 1)
        assert pl == (k - 1) // 2 * dil
    g = torch.Generator().manual_seed(5)
    for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
        w, b, pl = _dump_cl(lib, h, blob, 1, i)
        cin = hp.upsample_initial_channel >> i
        cout = cin // 2
        assert w.shape[0] == u * cout and w.shape[1] == cin
        kk = w.shape[2]
        x = bf(torch.randn(2, cin, 37, generator=g))
        wt = bf(O.fold_weight_norm(sd, f"dec.ups.{i}"))
        ref = F.conv_transpose1d(x.double(), wt.double(), sd[f"dec.ups.{i}.bias"].double(), stride=u, padding=(k - u) // 2)
        y = F.conv1d(F.pad(x.double(), (pl, kk - 1 - pl)), w.double(), b.double())          # [B, u*cout, L]
        y = y.view(2, u, cout, 37).permute(0, 2, 3, 1).reshape(2, cout, 37 * u)            # out[t*u+ph][co] = y[ph*cout+co][t]
        assert ref.shape == y.shape
        assert (ref - y).abs().max().item() < 1e-3 * ref.abs().max().item(), f"ups {i}"   # rare 1-ulp bf16 flips of the fold
    assert lib.bv2_set_generator_dtype(h, L.BF16) == 0 and lib.bv2_set_generator_dtype(h, L.F32) == 0
    assert lib.bv2_set_generator_dtype(h, 1) != 0


def test_bf16_oracle_generator_close_to_fp32():
    """The bf16-storage restatement (oracle generator_bf16) stays within bf16 round-off of the fp32 Generator."""
    hp = H.default_v23()
    sd = cached_state_dict(hp, 0)
    g = torch.Generator().manual_seed(3)
    z = torch.randn(1, hp.inter_channels, 6, generator=g)
    gg = torch.randn(1, hp.gin_channels, 1, generator=g)
    with torch.no_grad():
        a = O.generator(sd, hp, z, gg)
        b = O.generator_bf16(sd, hp, z, gg)
    rel = ((a - b).pow(2).mean().sqrt() / a.pow(2).mean().sqrt()).item()
    assert 1e-5 < rel < 5e-2, rel


def test_fp16_flow_streams_and_tap_major_resblock_streams():
    """The fp16 weight streams of the transformer flow's Encoder convs decode to fp16(reference weight) — for the fused q/k/v
    projection: q rows pre-scaled by 1/sqrt(d), k, v rows as is, then per head the 2W+1 relative-key rows E_k·W_q/sqrt(d)
    (attentions.py:264-266, 280, 286-288) — and the tap-major whole-ResBlock streams hold the same bf16 weights as the
    per-conv streams."""
    import math
    hp = H.default_v23()
    sd = cached_state_dict(hp, 0)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    blob = m.pack_host_blob()
    lib = L.load()
    lib.bv2_test_dump_cl_conv.restype = C.c_int
    h = m._handle
    h16 = lambda t: t.to(torch.float16).to(torch.float32)
    nf = hp.n_flow_layer
    for a in (0, nf - 1):                                   # application order a <-> reference flows.{2*(nf-1-a)}
        p = f"flow.flows.{2 * (nf - 1 - a)}.enc"
        for lyr in (0, hp.n_layers_trans_flow - 1):
            w, b, _ = _dump_cl(lib, h, blob, 3, a, lyr, 1)
            assert torch.equal(w, h16(sd[f"{p}.attn_layers.{lyr}.conv_o.weight"])) and torch.equal(b, sd[f"{p}.attn_layers.{lyr}.conv_o.bias"])
            w, b, pl = _dump_cl(lib, h, blob, 3, a, lyr, 2)
            assert pl == 2 and torch.equal(w, h16(sd[f"{p}.ffn_layers.{lyr}.conv_1.weight"]))
            w, b, _ = _dump_cl(lib, h, blob, 3, a, lyr, 3)
            assert torch.equal(w, h16(sd[f"{p}.ffn_layers.{lyr}.conv_2.weight"])) and torch.equal(b, sd[f"{p}.ffn_layers.{lyr}.conv_2.bias"])
            w, b, _ = _dump_cl(lib, h, blob, 3, a, lyr, 0)
            Hc, dk, nr = hp.hidden_channels, hp.hidden_channels // hp.n_heads, 9
            assert w.shape == (3 * Hc + hp.n_heads * nr, Hc, 1)
            wq = sd[f"{p}.attn_layers.{lyr}.conv_q.weight"][:, :, 0].double()
            isq = 1.0 / math.sqrt(dk)
            assert torch.equal(w[:Hc, :, 0], h16((wq * isq).float()))
            assert torch.equal(w[Hc:2 * Hc], h16(sd[f"{p}.attn_layers.{lyr}.conv_k.weight"]))
            assert torch.equal(w[2 * Hc:3 * Hc], h16(sd[f"{p}.attn_layers.{lyr}.conv_v.weight"]))
            ek = sd[f"{p}.attn_layers.{lyr}.emb_rel_k"][0].double()
            rel = torch.cat([ek @ wq[hh * dk:(hh + 1) * dk] for hh in range(hp.n_heads)]) * isq
            assert (w[3 * Hc:, :, 0].double() - rel).abs().max() <= 2.0 ** -10 * rel.abs().max()   # one fp16 rounding
    for i in (3, 4):                                        # the narrow stages carry a whole-ResBlock stream
        for j in range(len(hp.resblock_kernel_sizes)):
            for d in range(3):
                for e in (0, 1):
                    w2, b2, pl2 = _dump_cl(lib, h, blob, 2, i, j, d, e)
                    w4, b4, pl4 = _dump_cl(lib, h, blob, 4, i, j, d, e)
                    assert pl2 == pl4 and torch.equal(w2, w4) and torch.equal(b2, b4)
                    if i == 4:                                   # C = 16: the tap-pair stream of resblock_c16_bf16.hip holds the same values
                        w5, b5, pl5 = _dump_cl(lib, h, blob, 5, i, j, d, e)
                        assert pl2 == pl5 and torch.equal(w2, w5) and torch.equal(b2, b5)
    dims = (C.c_int32 * 4)()
    assert lib.bv2_test_dump_cl_conv(h, C.c_void_p(blob.data_ptr()), 4, 0, 0, 0, 0, dims, None, None) == -2   # wide stage: none
    assert lib.bv2_test_dump_cl_conv(h, C.c_void_p(blob.data_ptr()), 5, 3, 0, 0, 0, dims, None, None) == -2   # C = 32: no tap-pair stream


def test_no_kernel_on_the_default_path_has_a_scratch_segment():
    """A kernel that spills registers gets a scratch segment, and a dispatch with a scratch segment drains the queue on this stack: one
    such kernel (2 spilled registers, 6 launches per step) cost +0.3 ms per 4.6 ms step in round 3 while the sum of kernel times fell.
    tools/kernel_resources.py --check compiles every kernel source for gfx950 (no GPU needed) and fails on ScratchSize > 0 outside
    the few variants the default model never launches."""
    import subprocess
    import sys
    from tests.helpers import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_resources.py"), "--check"], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]


def test_x6_split_is_exact_and_round_to_nearest():
    """The packer's three-way bf16 split (bv2_kernels.h x6_split, behind the x6 weight planes): h1 + h2 + h3 == v EXACTLY for every
    finite fp32 whose low planes stay normal, and each plane is the round-to-nearest-even bf16 of what is left."""
    import ctypes as C
    import numpy as np
    from bert_vits2_amd import lib as L
    lib = L.load()
    lib.bv2_test_x6_split.restype = None
    lib.bv2_test_x6_split.argtypes = [C.c_float, C.POINTER(C.c_uint16)]
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(2000).astype(np.float32) * np.exp(rng.uniform(-30, 30, 2000)).astype(np.float32),
                           np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1e-30, np.float32(1) + np.float32(2 ** -23),
                                     np.float32(1) - np.float32(2 ** -24), 0.1, 1 / 3], dtype=np.float32)])
    h = (C.c_uint16 * 3)()

    def bf(u):
        return np.array([int(u) << 16], dtype=np.uint32).view(np.float32)[0]

    def rne(v):
        u = int(np.array([v], dtype=np.float32).view(np.uint32)[0])
        u += 0x7fff + ((u >> 16) & 1)
        return (u >> 16) & 0xffff

    for v in vals:
        lib.bv2_test_x6_split(float(v), h)
        planes = [bf(h[i]) for i in range(3)]
        assert float(np.float64(planes[0]) + np.float64(planes[1]) + np.float64(planes[2])) == float(v), (v, planes)
        rest = np.float32(v)
        for i in range(3):
            assert h[i] == rne(rest), (v, i)
            rest = np.float32(rest - planes[i])
    # the top of the range: plane 1 saturates at the largest bf16 (RNE would give inf and a NaN remainder); still exact, all finite
    fmax = np.finfo(np.float32).max
    with np.errstate(over="ignore", invalid="ignore"):
        for v in (fmax, -fmax, np.float32(3.3962e38), np.float32(-3.3962e38), np.float32(3.3961e38), np.nextafter(np.float32(bf(0x7f7f)), np.float32(np.inf))):
            lib.bv2_test_x6_split(float(v), h)
            assert (h[0] & 0x7fff) == 0x7f7f and (h[0] >> 15) == int(v < 0), (v, hex(h[0]))
            planes = [bf(h[i]) for i in range(3)]
            assert all(np.isfinite(p) for p in planes)
            assert float(np.float64(planes[0]) + np.float64(planes[1]) + np.float64(planes[2])) == float(v), (v, planes)
        lib.bv2_test_x6_split(float(bf(0x7f7f)), h)                       # the largest bf16 itself: unchanged
        assert (h[0], h[1], h[2]) == (0x7f7f, 0, 0)
        lib.bv2_test_x6_split(float("inf"), h)                           # non-finite stays non-finite
        assert h[0] == 0x7f7f and not np.isfinite(bf(h[1]))
        lib.bv2_test_x6_split(float("nan"), h)
        assert np.isnan(bf(h[0]))


def test_resblock2_config_packs_and_the_shorter_config_struct_still_means_resblock1():
    """`resblock: "2"` (reference models.py:508, modules.py:318-363): the schema names `dec.resblocks.N.convs.{0,1}`, the packer takes them
    (no CPU compute here), and a caller built against the pre-round-5 bv2_config (no trailing resblock_type) is still accepted."""
    lib = L.load()
    hp = H.default_v23(resblock="2", resblock_kernel_sizes=(3, 5, 7), resblock_dilation_sizes=((1, 2), (2, 6), (3, 12)))
    names = schema.inference_schema(hp) if hasattr(schema, "inference_schema") else None
    m = models.from_hparams(hp)
    keys = set(m.state_dict().keys())
    assert "dec.resblocks.0.convs.1.weight_v" in keys and "dec.resblocks.14.convs.0.weight_g" in keys
    assert not any(".convs1." in k or ".convs2." in k for k in keys)
    from bert_vits2_amd import synth
    m.load_state_dict(synth.synthetic_state_dict(hp, seed=1), strict=False)
    blob = m.pack_host_blob()                                       # bv2_create + bv2_load_tensor x N + bv2_pack_weights
    assert blob.numel() > 1 << 20
    cfg = L.make_config(hp)
    assert cfg.resblock_type == 2 and cfg.n_resblock_dilations == 2
    with pytest.raises(ValueError):
        H.default_v23(resblock="2", resblock_dilation_sizes=((1,), (1,), (1,))).validate()
    # the shorter struct: everything up to (not including) resblock_type
    cfg1 = L.make_config(H.default_v23())
    cfg1.struct_bytes = C.sizeof(L.Config) - 4
    h = C.c_void_p()
    assert lib.bv2_create(C.byref(cfg1), C.byref(h)) == 0, lib.bv2_last_error(None)
    n_short = lib.bv2_packed_bytes(h)
    lib.bv2_destroy(h)
    cfg1 = L.make_config(H.default_v23())
    assert lib.bv2_create(C.byref(cfg1), C.byref(h)) == 0
    assert lib.bv2_packed_bytes(h) == n_short                        # same layout either way
    lib.bv2_destroy(h)


def test_every_option_key_is_documented_and_every_documented_key_exists():
    """`bv2_set_option` keys (csrc/bv2_api.cpp) against the option list in include/bv2.h's comment: a switch nobody can find in the header, or
    a documented one the library rejects, is a boundary bug.  ("bv2_bert_set_option" keys live in include/bv2_bert.h.)"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    api = open(os.path.join(root, "bert-vits2_amd", "csrc", "bv2_api.cpp")).read()
    body = api[api.index("int bv2_set_option("):]
    body = body[:body.index("unknown key")]
    keys = set(re.findall(r'k == "([a-z0-9_]+)"', body))
    hdr = open(os.path.join(root, "include", "bv2.h")).read()
    documented = set(re.findall(r'^ \*   "([a-z0-9_]+)"', hdr, flags=re.M))
    assert len(keys) >= 20
    assert keys - documented == set(), f"undocumented option keys: {sorted(keys - documented)}"
    assert documented - keys == set(), f"documented but unknown to bv2_set_option: {sorted(documented - keys)}"
