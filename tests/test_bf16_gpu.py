"""GPU: the bf16 Generator (BASELINE config 3: bf16 weights / channels-last bf16 activations, fp32 accumulate;
kernels/gen_bf16.hip) — single kernel against a torch reference with the SAME rounding points, then the whole Generator
and infer() against the bf16-storage oracle (oracle generator_bf16), the fp32 oracle and the reference golden outputs.

Tolerances.  The kernel reference shares every rounding point with the kernel, so the only difference is fp32 summation
order: outputs agree except for rare 1-ulp bf16 flips (2^-7 relative).  Over the 30+ layers of the Generator those flips
propagate, so the whole-Generator bar against the bf16 oracle is a RELATIVE waveform RMS error <= 1e-2 (expected ~2e-3),
and against the fp32 reference/golden <= 5e-2 (the reference's own bf16-autocast run differs from its fp32
run by ~2.5 % relative), mel-L1 reported.  Durations / path stay exact: encoder, durations and flow remain fp32."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import bv2_oracle as O, cases, mel
from tests.helpers import cached_state_dict, load_golden, rms, valid_wave_mask

pytestmark = pytest.mark.gpu


def _lib():
    from bert_vits2_amd import lib as L
    lib = L.load()
    lib.bv2_test_conv_cl_pack_bytes.restype = C.c_int64
    lib.bv2_test_conv_cl_pack_bytes.argtypes = [C.c_int] * 3
    lib.bv2_test_conv_cl_bf16.restype = C.c_int
    lib.bv2_test_conv_cl_bf16.argtypes = ([C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 6 + [C.c_int] * 8 + [C.c_float])
    return lib


def P(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def bf(t):
    return t.to(torch.bfloat16).to(torch.float32)


CL_CASES = [
    # B, cin, cout, k, dil, L, nsrc, lrelu, res, bias2, pad_left      (variant by cout: >=256 8x1, 96..255 4x1, 64 2x2, <=32 1x4)
    (1, 16, 16, 3, 1, 700, 1, 1, 0, 0, -1),
    (2, 16, 16, 11, 5, 1300, 1, 1, 1, 0, -1),
    (2, 32, 32, 7, 3, 777, 1, 1, 1, 0, -1),
    (1, 32, 16, 1, 1, 515, 3, 1, 0, 0, -1),          # ups[4]-like: mean of 3 branches
    (2, 64, 64, 11, 1, 600, 1, 1, 1, 0, -1),
    (1, 64, 64, 3, 5, 257, 1, 0, 0, 0, -1),          # raw pass-through staging
    (2, 128, 128, 11, 5, 300, 1, 1, 1, 0, -1),
    (1, 128, 128, 5, 1, 129, 3, 1, 0, 0, 2),         # ups[2]-like window
    (1, 256, 256, 7, 3, 200, 1, 1, 1, 0, -1),
    (2, 192, 512, 7, 1, 50, 1, 0, 0, 1, -1),         # conv_pre: per-batch bias
    (1, 512, 2048, 3, 1, 40, 1, 1, 0, 0, 1),         # ups[0] in channels-last form
    (1, 48, 96, 3, 2, 131, 2, 1, 0, 0, -1),          # odd widths
]


@pytest.mark.parametrize("B,cin,cout,k,dil,L,nsrc,lrelu,res,bias2,pad_left", CL_CASES)
def test_conv_cl_bf16_kernel(B, cin, cout, k, dil, L, nsrc, lrelu, res, bias2, pad_left):
    lib = _lib()
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + k + L)
    xs = [bf(torch.randn(B, cin, L, generator=g)) for _ in range(nsrc)]             # values exactly representable in bf16
    w = torch.randn(cout, cin, k, generator=g) / math.sqrt(cin * k)
    bias = torch.randn(cout, generator=g)
    b2 = torch.randn(B, cout, generator=g) if bias2 else None
    r = bf(torch.randn(B, cout, L, generator=g)) if res else None
    # reference with the kernel's rounding points (bv2_kernels.h ClProb)
    xin = xs[0]
    if nsrc > 1:                                   # the stage hand-over's rounding points (kernels/cl_bf16.h stage_mean): widest branch first, running sum in bf16
        acc = xs[-1]
        for t in xs[-2:0:-1]:
            acc = bf(acc + t)
        xin = bf((acc + xs[0]) * torch.tensor(1.0 / nsrc, dtype=torch.float32))
    if lrelu:
        xin = torch.where(xin < 0, xin * torch.tensor(0.1, dtype=torch.float32), xin)
    if lrelu or nsrc > 1:
        xin = bf(xin)
    pl = (k - 1) // 2 * dil if pad_left < 0 else pad_left
    pr = (k - 1) * dil - pl
    ref = F.conv1d(F.pad(xin.double(), (pl, pr)), bf(w).double(), bias.double(), dilation=dil)
    if b2 is not None:
        ref = ref + b2.double()[:, :, None]
    if r is not None:
        ref = ref + r.double()
    cl = lambda t: t.transpose(1, 2).contiguous().to(torch.bfloat16).cuda()           # [B][L][C] bf16
    xd = [cl(t) for t in xs] + [None, None]
    out = torch.full((B, L, cout), float("nan"), dtype=torch.bfloat16, device="cuda")
    rd = cl(r) if r is not None else None
    b2d = b2.cuda() if b2 is not None else None
    wp = torch.empty(lib.bv2_test_conv_cl_pack_bytes(cin, cout, k), dtype=torch.uint8, device="cuda")
    rc = lib.bv2_test_conv_cl_bf16(None, P(xd[0]), P(xd[1]), P(xd[2]), nsrc, P(w), P(bias), P(wp), P(out), P(rd), P(b2d),
                                   B, cin, cout, k, dil, pad_left, L, lrelu, 0.1)
    assert rc == 0
    torch.cuda.synchronize()
    got = out.float().cpu().transpose(1, 2).double()
    assert torch.isfinite(got).all()
    scale = ref.abs().max().item()
    err = (got - ref).abs()
    # output = bf16(fp32 sum): half an ulp of rounding (2^-9 relative) + summation-order noise
    tol = 2.0 ** -8 * ref.abs() + 2e-5 * scale
    assert bool((err <= tol).all()), (err.max().item(), scale, (err > tol).float().mean().item())


def _gpu_model(hp, seed):
    from bert_vits2_amd import models
    m = models.from_hparams(hp)
    m.load_state_dict(cached_state_dict(hp, seed), strict=False)
    return m.to("cuda").eval()


def _relrms(a, b):
    return rms(a - b) / max(rms(b), 1e-30)


@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged", "rb2_b2_t14", "narrow_b2_t18"])      # rb2: `resblock: "2"` (modules.ResBlock2), layer-wise; narrow: three stages 8 x 4 x 2, final width 32
def test_stage_generator_bf16_vs_bf16_oracle(name):
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    ref32 = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                    batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, want_taps=True, **kw)
    zin = (ref32["z"] * ref32["y_mask"])
    taps16 = {}
    with torch.no_grad():
        o16 = O.generator_bf16(sd, hp, zin, ref32["g"], None, taps16)
    m = _gpu_model(hp, seed)
    m.set_generator_dtype(torch.bfloat16)
    B, Cc, Ty = ref32["z"].shape
    taps = {}
    up = 1
    for i, u in enumerate(hp.upsample_rates):
        up *= u
        ch = hp.upsample_initial_channel // 2 ** (i + 1)
        taps[f"dec.ups.{i}"] = torch.zeros(B, ch, Ty * up, device="cuda")
        for j in range(3):
            taps[f"dec.rb.{i}.{j}"] = torch.zeros(B, ch, Ty * up, device="cuda")
    for k, t in taps.items():
        m.set_tap(k, t)
    try:
        o = m.stage_generator(ref32["z"], ref32["y_lengths"], ref32["g"])
        torch.cuda.synchronize()
    finally:
        m.set_tap(None)
    report = []
    for i in range(len(hp.upsample_rates)):
        report.append((f"ups{i}", _relrms(taps[f"dec.ups.{i}"].cpu(), taps16[f"dec.ups.{i}"])))
        stage = (taps[f"dec.rb.{i}.0"] + taps[f"dec.rb.{i}.1"] + taps[f"dec.rb.{i}.2"]).cpu() * torch.tensor(1.0 / 3)
        report.append((f"stage{i}", _relrms(stage, taps16[f"dec.stage.{i}"])))
    e16, e32 = _relrms(o.cpu(), o16), _relrms(o.cpu(), ref32["o"])
    print(f"\n[{name}] bf16 Generator: rel RMS vs bf16 oracle {e16:.3e}, vs fp32 oracle {e32:.3e}; per stage {report}")
    assert torch.isfinite(o).all()
    for tag, e in report:
        assert e < 1e-2, report
    assert e16 < 1e-2, e16
    assert e32 < 5e-2, e32
    # the fp32 path is untouched by the switch
    m.set_generator_dtype(torch.float32)
    o_f = m.stage_generator(ref32["z"], ref32["y_lengths"], ref32["g"])
    assert rms(o_f.cpu() - ref32["o"]) < 2e-5


def test_infer_bf16_generator_end_to_end_vs_reference_golden():
    name = "mix_b2_ragged"
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    meta, gold = load_golden(name)
    m = _gpu_model(hp, seed)
    m.set_generator_dtype(torch.bfloat16)
    args = (batch["x"].cuda(), batch["x_lengths"].cuda(), batch["sid"].cuda(), batch["tone"].cuda(), batch["language"].cuda(),
            batch["bert"].cuda(), batch["ja_bert"].cuda(), batch["en_bert"].cuda())
    o, attn, y_mask, (z, z_p, m_p, logs_p) = m.infer(*args, noise_w=nw, noise_z=nz.cuda(), w_ceil=gold["w_ceil"], **kw)
    torch.cuda.synchronize()
    assert torch.equal(attn.cpu(), gold["attn"])               # durations / path: fp32 front end, exact
    assert o.shape == gold["o"].shape and o.dtype == torch.float32
    S = o.shape[2]
    vm = valid_wave_mask(gold["y_lengths"], hp.total_upsample, S).expand_as(gold["o"])
    d = (o.cpu() - gold["o"])[vm]
    rel = rms(d) / rms(gold["o"][vm])
    ml = mel.mel_l1(o.cpu()[:, 0].numpy(), gold["o"][:, 0].numpy(), gold["y_lengths"].numpy() * hp.total_upsample)
    print(f"\n[{name}] bf16 Generator end to end vs REFERENCE golden: rel RMS {rel:.3e} (abs {rms(d):.3e}), mel-L1 {ml:.3e}")
    assert rel < 5e-2, rel
    assert maxrel_z(z, gold["z"]) < 1e-4                        # flow output is still the fp32 path


def maxrel_z(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-20)).item()


@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged", "narrow_b2_t18"])      # narrow: rates 8 / 4 / 2 with kernels 16 / 8 / 4
def test_convtranspose_phase_taps_bit_identical(name):
    """Round 5: a bf16 ConvTranspose1d launch (one conv over the union of the u phases' tap windows, reference models.py:538-549) runs per wave
    only the taps of the phases its output channels belong to ("ups_phase_taps", default 1).  The taps it steps over are zeros the packer
    wrote, so every stage's output — and the waveform — must be bit-identical to the run that multiplies through them (= 0)."""
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    sd = cached_state_dict(hp, seed)
    ref32 = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                    batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, want_taps=True, **kw)
    m = _gpu_model(hp, seed)
    m.set_generator_dtype(torch.bfloat16)
    B, Cc, Ty = ref32["z"].shape
    got = {}
    for v in (1, 0):
        m.set_option("ups_phase_taps", v)
        taps = {}
        up = 1
        for i, u in enumerate(hp.upsample_rates):
            up *= u
            taps[f"dec.ups.{i}"] = torch.zeros(B, hp.upsample_initial_channel // 2 ** (i + 1), Ty * up, device="cuda")
        for k, t in taps.items():
            m.set_tap(k, t)
        try:
            o = m.stage_generator(ref32["z"], ref32["y_lengths"], ref32["g"])
            torch.cuda.synchronize()
        finally:
            m.set_tap(None)
        got[v] = (o.cpu(), {k: t.cpu() for k, t in taps.items()})
    m.set_option("ups_phase_taps", 1)
    for k in got[1][1]:
        assert torch.equal(got[1][1][k], got[0][1][k]), k
        assert float(got[1][1][k].abs().max()) > 0
    assert torch.equal(got[1][0], got[0][0])


@pytest.mark.parametrize("name", ["zh_b1_t24", "mix_b2_ragged"])
def test_conv_post_rowwise_kernel_vs_any_width_kernel(name):
    """Round 5: conv_post + tanh of the bf16 path (reference models.py:553-555) at C = 16, k = 7 on the row-wise kernel (default) against the
    any-width kernel ("conv_post_rows" = 0): the same fp32 products summed in another order -> a few ulp on a waveform in [-1, 1]; ragged
    lengths included (rows past an utterance's end are the conv's zero padding in both)."""
    hp, seed, batch, nw, nz, kw = cases.build_case(name)
    assert hp.upsample_initial_channel >> len(hp.upsample_rates) == 16
    sd = cached_state_dict(hp, seed)
    ref32 = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                    batch["ja_bert"], batch["en_bert"], noise_w=nw, noise_z=nz, want_taps=True, **kw)
    m = _gpu_model(hp, seed)
    m.set_generator_dtype(torch.bfloat16)
    outs = {}
    for v in (1, 0):
        m.set_option("conv_post_rows", v)
        outs[v] = m.stage_generator(ref32["z"], ref32["y_lengths"], ref32["g"]).cpu()
    m.set_option("conv_post_rows", 1)
    d = (outs[1] - outs[0]).abs().max().item()
    print(f"\n[{name}] conv_post row-wise vs any-width: max |delta| {d:.3e} (signal rms {rms(outs[0]):.3e})")
    assert rms(outs[0]) > 0 and d < 2e-6
    assert not torch.equal(outs[1], outs[0]) or d == 0.0
