"""GPU: the drop-in boundary on the device —
 * the one-shot C entry point ``bv2_infer`` (incl. its ``-3`` "T_y exceeds Ty_cap" contract) against the two-phase path;
 * a checkpoint FILE in the reference's ``utils.save_checkpoint`` format (utils.py:123-139) -> ``checkpoint.load_checkpoint`` ->
   ``infer()`` against the oracle, and the packed-blob cache round trip;
 * the RNG contract: a seeded run without injected noise draws the SDP noise from the CPU generator exactly as the
   reference does (durations equal the seeded REFERENCE fixture's) and takes the prior noise with the reference's strides;
 * handle hygiene after ``.to()``: ``decode()`` with an old ``enc`` repacks instead of reading freed memory."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from bert_vits2_amd import checkpoint, hparams as H, lib as L, models
from oracle import bv2_oracle as O, cases
from tests.helpers import GOLDEN, cached_state_dict, rms, valid_wave_mask

pytestmark = pytest.mark.gpu


def _model(hp, seed=0):
    m = models.from_hparams(hp)
    m.load_state_dict(cached_state_dict(hp, seed), strict=False)
    return m.to("cuda").eval()


def _args(b):
    return tuple(b[k].cuda() for k in ("x", "x_lengths", "sid", "tone", "language", "bert", "ja_bert", "en_bert"))


def test_bv2_infer_one_shot_matches_two_phase_and_reports_ty_cap():
    hp, seed, batch, nw, nz, kw = cases.build_case("mix_b2_ragged")
    m = _model(hp, seed)
    o_ref, attn_ref, ym_ref, (z_ref, zp_ref, mp_ref, lp_ref) = m.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), **kw)
    torch.cuda.synchronize()
    B, T = batch["x"].shape
    Ty = ym_ref.shape[2]
    lib, h = m._lib, m._handle
    dev = "cuda"
    a = _args(batch)
    nwd, nzd = nw.cuda().contiguous(), nz.cuda().contiguous()
    e = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    eo = dict(g=e(B, hp.gin_channels), x=e(B, hp.hidden_channels, T), m_p=e(B, hp.inter_channels, T), logs_p=e(B, hp.inter_channels, T),
              x_mask=e(B, T), logw_sdp=e(B, T), logw_dp=e(B, T), logw=e(B, T), w_ceil=e(B, T),
              y_lengths=torch.empty(B, dtype=torch.int64, device=dev))
    P = lambda t: C.c_void_p(t.data_ptr())
    ein = L.EncodeIn(B, T, P(a[0]), P(a[1]), P(a[2]), P(a[3]), P(a[4]), P(a[5]), P(a[6]), P(a[7]), P(nwd), kw["noise_scale_w"],
                     kw["sdp_ratio"], kw["length_scale"])
    eout = L.EncodeOut(*[P(eo[k]) for k in ("g", "x", "m_p", "logs_p", "x_mask", "logw_sdp", "logw_dp", "logw", "w_ceil", "y_lengths")])
    cap = Ty + 40
    S = cap * hp.total_upsample
    do = dict(o=e(B, 1, S), attn=e(B, 1, cap, T), y_mask=e(B, 1, cap), z=e(B, hp.inter_channels, cap), z_p=e(B, hp.inter_channels, cap),
              m_p=e(B, hp.inter_channels, cap), logs_p=e(B, hp.inter_channels, cap))
    dout = L.DecodeOut(*[P(do[k]) for k in ("o", "attn", "y_mask", "z", "z_p", "m_p", "logs_p")])
    nbytes = lib.bv2_workspace_bytes(h, B, T, cap)
    ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dev)
    ty_out = C.c_int32(0)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.bv2_infer(h, stream, C.byref(ein), C.byref(eout), P(nzd), nzd.stride(0), nzd.stride(1), nzd.stride(2), kw["noise_scale"], 0,
                       cap, C.byref(dout), C.byref(ty_out), P(ws), ws.numel())
    torch.cuda.synchronize()
    assert rc == 0, lib.bv2_last_error(h)
    assert ty_out.value == Ty
    # outputs are written densely with the REALISED T_y as row stride
    o = do["o"].flatten()[: B * Ty * hp.total_upsample].view(B, 1, -1)
    assert torch.equal(o, o_ref)
    assert torch.equal(do["z"].flatten()[: B * hp.inter_channels * Ty].view(B, -1, Ty), z_ref)
    assert torch.equal(do["attn"].flatten()[: B * Ty * T].view(B, 1, Ty, T), attn_ref)
    assert torch.equal(eo["w_ceil"], m.last_encode["w_ceil"])
    # a cap below the realised T_y: -3, *Ty_out still tells the caller how much to allocate
    ty_out = C.c_int32(0)
    rc = lib.bv2_infer(h, stream, C.byref(ein), C.byref(eout), P(nzd), nzd.stride(0), nzd.stride(1), nzd.stride(2), kw["noise_scale"], 0,
                       Ty - 1, C.byref(dout), C.byref(ty_out), P(ws), ws.numel())
    assert rc == -3 and ty_out.value == Ty and b"Ty_cap" in lib.bv2_last_error(h)


def test_checkpoint_file_to_infer_vs_oracle_and_packed_cache(tmp_path):
    hp, seed, batch, nw, nz, kw = cases.build_case("zh_b1_t24")
    sd = cached_state_dict(hp, seed)
    full = dict(sd)
    full["enc_q.pre.weight"] = torch.zeros(192, 1025, 1)                  # a real G_*.pth also carries the training-only nets
    path = tmp_path / "G_8000.pth"
    torch.save(dict(model=full, iteration=8000, optimizer=None, learning_rate=1.5e-4), path)       # utils.save_checkpoint's dict
    m = models.from_hparams(hp).to("cuda").eval()
    out = checkpoint.load_checkpoint(str(path), m, None, skip_optimizer=True)
    assert out[0] is m and out[3] == 8000 and m.last_missing_keys == []
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, **kw)
    o, attn, y_mask, _ = m.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)
    torch.cuda.synchronize()
    assert torch.equal(attn.cpu(), ref["attn"])
    assert rms(o.cpu() - ref["o"]) <= 5e-5
    # packed-blob cache: save -> a fresh model that never saw the parameters attaches it and produces the same audio
    cache = tmp_path / "G_8000.bv2"
    n = checkpoint.save_packed(m, str(cache))
    assert n == os.path.getsize(cache)
    m2 = models.from_hparams(hp)
    checkpoint.load_packed(m2, str(cache), device="cuda")
    o2, *_ = m2.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), w_ceil=ref["w_ceil"], **kw)
    assert torch.equal(o2, o)
    # a blob whose pack-layout stamp differs (a cache written by an older build) is refused
    raw = bytearray(open(cache, "rb").read())
    raw[12:16] = (9999).to_bytes(4, "little")
    stale = tmp_path / "stale.bv2"
    open(stale, "wb").write(raw)
    with pytest.raises(RuntimeError, match="pack layout"):
        checkpoint.load_packed(models.from_hparams(hp), str(stale), device="cuda")
    # a blob-only model cannot repack after .to(): it says so instead of packing placeholder weights
    m2.to("cuda")
    with pytest.raises(RuntimeError, match="no weights"):
        m2.infer(*_args(batch), noise_w=nw, noise_z=nz.cuda(), **kw)


def test_seeded_run_honours_the_reference_rng_contract():
    z = np.load(os.path.join(GOLDEN, "seeded_" + cases.SEEDED_CASE + ".npz"), allow_pickle=False)
    gold = {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}
    hp, seed, batch, _nw, _nz, kw = cases.build_case(cases.SEEDED_CASE)
    m = _model(hp, seed)
    torch.manual_seed(cases.SEEDED_SEED)
    o, attn, y_mask, (zz, z_p, m_p, logs_p) = m.infer(*_args(batch), **kw)
    torch.cuda.synchronize()
    # draw #1 comes from the CPU generator exactly as in the reference (models.py:248-251): the durations of the seeded REFERENCE run
    w_gold = gold["attn"][:, 0].sum(1)                                    # [B,T] frames per symbol
    flips = (m.last_encode["w_ceil"].cpu() != w_gold).float().mean().item()
    assert flips <= 0.01, flips
    if flips == 0:
        assert torch.equal(y_mask.cpu(), gold["y_mask"]) and torch.equal(attn.cpu(), gold["attn"])
    # draw #2: the recipe the shim uses is reproducible from the same seed and is consumed with the reference's strides
    torch.manual_seed(cases.SEEDED_SEED)
    B, T = batch["x"].shape
    nw = models.draw_noise_w(B, T, "cuda")
    assert torch.equal(nw.cpu(), gold["noise_w"])
    nzd = models.draw_noise_z(B, hp.inter_channels, y_mask.shape[2], "cuda")
    assert tuple(nzd.stride()) == (hp.inter_channels * y_mask.shape[2], 1, hp.inter_channels)
    o2, *_ = m.infer(*_args(batch), noise_w=nw, noise_z=nzd, **kw)
    assert torch.equal(o2, o)
    # and the strided tensor is read element-for-element like its contiguous copy
    o3, *_ = m.infer(*_args(batch), noise_w=nw, noise_z=nzd.contiguous(), **kw)
    assert torch.equal(o3, o)
    eps = (z_p - m_p) / torch.exp(logs_p) / kw["noise_scale"]
    valid = y_mask.bool().expand_as(eps)
    assert abs(eps[valid].mean().item()) < 0.05 and abs(eps[valid].std().item() - 1.0) < 0.05


def test_decode_after_to_repacks_instead_of_reading_freed_memory():
    hp, seed, batch, nw, nz, kw = cases.build_case("zh_b1_t24")
    m = _model(hp, seed)
    enc = m.encode_durations(*_args(batch), nw, noise_scale_w=kw["noise_scale_w"], sdp_ratio=kw["sdp_ratio"], length_scale=kw["length_scale"])
    Ty = int(enc["y_lengths"].max())
    d1 = m.decode(enc, nz.cuda(), Ty, noise_scale=kw["noise_scale"])
    m.to("cuda")                                                            # invalidates the packed blob and the C handle's pointer
    assert m._blob is None
    d2 = m.decode(enc, nz.cuda(), Ty, noise_scale=kw["noise_scale"])
    assert torch.equal(d1["o"], d2["o"])
