"""CPU restatement of reference ``SynthesizerTrn.infer`` — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / the timed CPU baseline.  The product path
(``bert-vits2_amd``) never imports it and fails loudly when its HIP library is missing.

What it is: a functional (no nn.Module), torch-CPU fp32/fp64 restatement of the algorithm
of reference models.py:1026-1074 and everything it calls, written from the reference's
semantics, working directly on a reference-schema ``state_dict``.
It deliberately uses a *different formulation* from the reference where that is the
natural way to say the same maths (banded relative-position attention instead of the
pad/reshape skew trick; gather-based length regulation instead of the one-hot matmul;
channel flips kept explicit), which is what makes agreeing with the real reference a
meaningful check.

Pinning: the reference ships NO tests / golden vectors / KATs for this path (SURVEY.md §4,
§8c: "parity unpinned by the reference").  This restatement is pinned instead against outputs of
the reference's own code run in the build container: ``oracle/gen_golden.py`` imports
``/root/reference/models.py`` unmodified, loads the same seeded synthetic checkpoint, injects the same
noise, and stores the reference outputs under ``tests/golden/``;
``tests/test_oracle_vs_golden.py`` holds this file to those outputs (fp32 round-off level).

Every function cites the reference file:line it follows.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.1          # reference modules.py:14
WINDOW = 4                 # reference attentions.py:46
COND_LAYER = 2             # reference attentions.py:69-71


# --------------------------------------------------------------------------------------------
# small helpers

def sequence_mask(lengths: torch.Tensor, max_len: int) -> torch.Tensor:
    """reference commons.py:119-123 → bool [B, max_len]."""
    return torch.arange(max_len, dtype=lengths.dtype)[None, :] < lengths[:, None]


def fold_weight_norm(sd: Dict[str, torch.Tensor], prefix: str) -> torch.Tensor:
    """w = g * v / ||v||, norm over every dim but 0 (torch weight_norm dim=0; for ConvTranspose1d dim 0 is C_in:
    reference models.py:510-522).  Accepts already-folded checkpoints (``.weight``), as
    produced by reference Generator.remove_weight_norm (models.py:559-564)."""
    if prefix + ".weight" in sd:
        return sd[prefix + ".weight"]
    v, g = sd[prefix + ".weight_v"], sd[prefix + ".weight_g"]
    n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
    return v * (g / n)


def channel_layer_norm(x, gamma, beta, eps=1e-5):
    """reference modules.py:26-29 / attentions.py:21-24: LayerNorm over the channel dim of [B,C,T]."""
    return F.layer_norm(x.transpose(1, 2), (x.shape[1],), gamma, beta, eps).transpose(1, 2)


def conv1x1(sd, p, x):
    return F.conv1d(x, sd[p + ".weight"], sd.get(p + ".bias"))


# --------------------------------------------------------------------------------------------
# attentions.py

def _h(x: torch.Tensor) -> torch.Tensor:
    """fp16 storage rounding (round-to-nearest-even), kept in fp32 for the arithmetic."""
    return x.to(torch.float16).to(torch.float32)


def rel_attention(sd, p, x, mask_bt, n_heads, half=False):
    """MultiHeadAttention with windowed relative positions, reference attentions.py:263-322 (+ helpers
    :324-395), in banded form: logits get q_i·E_k[j-i+4]/sqrt(d) for |j-i| <= 4, outputs get
    sum_r p[i,i+r]·E_v[r+4]; masked pairs are SET to -1e4 (attentions.py:297).

    half=True pins the rounding points of the fp16 product path (bv2_set_flow_dtype): the projection inputs and
    weights are rounded to fp16 (q rows pre-scaled by 1/sqrt(d), the relative-key logits taken from the fused rows
    E_k·W_q/sqrt(d) — both rounded AFTER the fold, as the packer does), accumulation / bias / attention core fp32."""
    B, C, T = x.shape
    d = C // n_heads
    ek, ev = sd[p + ".emb_rel_k"][0], sd[p + ".emb_rel_v"][0]                      # [9,d]
    if half:
        isq = 1.0 / math.sqrt(d)
        xh = _h(x)
        wq, bq = sd[p + ".conv_q.weight"][:, :, 0].double(), sd[p + ".conv_q.bias"].double()
        qs = (F.conv1d(xh, _h((wq * isq).float())[:, :, None]) + (bq * isq).float()[None, :, None])
        qs = qs.view(B, n_heads, d, T).transpose(2, 3)
        k = (F.conv1d(xh, _h(sd[p + ".conv_k.weight"])) + sd[p + ".conv_k.bias"][None, :, None]).view(B, n_heads, d, T).transpose(2, 3)
        v = (F.conv1d(xh, _h(sd[p + ".conv_v.weight"])) + sd[p + ".conv_v.bias"][None, :, None]).view(B, n_heads, d, T).transpose(2, 3)
        wrel = torch.stack([ek.double() @ wq[hh * d:(hh + 1) * d] for hh in range(n_heads)]) * isq          # [H,9,C]
        brel = torch.stack([ek.double() @ bq[hh * d:(hh + 1) * d] for hh in range(n_heads)]) * isq          # [H,9]
        ql = F.conv1d(xh, _h(wrel.float()).reshape(n_heads * ek.shape[0], C, 1)) + brel.float().reshape(-1)[None, :, None]
        ql = ql.view(B, n_heads, ek.shape[0], T).transpose(2, 3)                   # [B,H,T,9]
    else:
        q = conv1x1(sd, p + ".conv_q", x).view(B, n_heads, d, T).transpose(2, 3)  # [B,H,T,d]
        k = conv1x1(sd, p + ".conv_k", x).view(B, n_heads, d, T).transpose(2, 3)
        v = conv1x1(sd, p + ".conv_v", x).view(B, n_heads, d, T).transpose(2, 3)
        qs = q / math.sqrt(d)
        ql = qs @ ek.t()                                                           # [B,H,T,9]
    scores = qs @ k.transpose(-1, -2)                                              # [B,H,T,T]
    idx = torch.arange(T)
    rel = idx[None, :] - idx[:, None]                                              # j - i
    band = rel.abs() <= WINDOW
    ridx = (rel + WINDOW).clamp(0, 2 * WINDOW)
    scores = scores + torch.where(band, ql.gather(-1, ridx.expand(B, n_heads, T, T)), torch.zeros(()))
    pair = (mask_bt[:, None, :, None] * mask_bt[:, None, None, :]) != 0
    scores = torch.where(pair, scores, torch.full((), -1e4, dtype=scores.dtype))
    pa = torch.softmax(scores, dim=-1)
    out = pa @ v                                                                   # [B,H,T,d]
    relw = torch.zeros(B, n_heads, T, 2 * WINDOW + 1, dtype=x.dtype)
    for r in range(-WINDOW, WINDOW + 1):
        lo, hi = max(0, -r), min(T, T - r)
        if hi > lo:
            i = torch.arange(lo, hi)
            relw[:, :, lo:hi, r + WINDOW] = pa[:, :, i, i + r]
    out = out + relw @ ev
    out = out.transpose(2, 3).reshape(B, C, T)
    if half:
        return F.conv1d(_h(out), _h(sd[p + ".conv_o.weight"]), sd[p + ".conv_o.bias"])
    return conv1x1(sd, p + ".conv_o", out)


def ffn(sd, p, x, mask, k, half=False):
    """reference attentions.py:438-464: mask, same-pad conv, ReLU, mask, same-pad conv, mask.
    half=True: conv inputs / weights / the hidden activation rounded to fp16, fp32 accumulation + bias."""
    pl, pr = (k - 1) // 2, k // 2
    if half:
        h = F.conv1d(F.pad(_h(x * mask), (pl, pr)), _h(sd[p + ".conv_1.weight"]), sd[p + ".conv_1.bias"])
        h = _h(torch.relu(h) * mask)
        h = F.conv1d(F.pad(h, (pl, pr)), _h(sd[p + ".conv_2.weight"]), sd[p + ".conv_2.bias"])
        return h * mask
    h = F.conv1d(F.pad(x * mask, (pl, pr)), sd[p + ".conv_1.weight"], sd[p + ".conv_1.bias"])
    h = torch.relu(h)
    h = F.conv1d(F.pad(h * mask, (pl, pr)), sd[p + ".conv_2.weight"], sd[p + ".conv_2.bias"])
    return h * mask


def encoder(sd, p, x, mask, g, n_layers, n_heads, ksize, half=False):
    """attentions.Encoder.forward, reference attentions.py:103-120. mask [B,1,T] float, g [B,gin,1]."""
    x = x * mask
    mbt = mask[:, 0, :]
    for i in range(n_layers):
        if i == COND_LAYER and g is not None:
            gg = F.linear(g.transpose(1, 2), sd[p + ".spk_emb_linear.weight"], sd[p + ".spk_emb_linear.bias"])
            x = (x + gg.transpose(1, 2)) * mask
        y = rel_attention(sd, f"{p}.attn_layers.{i}", x, mbt, n_heads, half)
        x = channel_layer_norm(x + y, sd[f"{p}.norm_layers_1.{i}.gamma"], sd[f"{p}.norm_layers_1.{i}.beta"])
        y = ffn(sd, f"{p}.ffn_layers.{i}", x, mask, ksize, half)
        x = channel_layer_norm(x + y, sd[f"{p}.norm_layers_2.{i}.gamma"], sd[f"{p}.norm_layers_2.{i}.beta"])
    return x * mask


# --------------------------------------------------------------------------------------------
# models.py: TextEncoder

def text_encoder(sd, hp, x, x_lengths, tone, language, bert, ja_bert, en_bert, g):
    """reference models.py:377-400."""
    T = x.shape[1]
    e = (F.embedding(x, sd["enc_p.emb.weight"]) + F.embedding(tone, sd["enc_p.tone_emb.weight"])
         + F.embedding(language, sd["enc_p.language_emb.weight"])
         + conv1x1(sd, "enc_p.bert_proj", bert).transpose(1, 2)
         + conv1x1(sd, "enc_p.ja_bert_proj", ja_bert).transpose(1, 2)
         + conv1x1(sd, "enc_p.en_bert_proj", en_bert).transpose(1, 2)) * math.sqrt(hp.hidden_channels)
    h = e.transpose(1, 2)
    mask = sequence_mask(x_lengths, T)[:, None, :].to(h.dtype)
    h = encoder(sd, "enc_p.encoder", h * mask, mask, g, hp.n_layers, hp.n_heads, hp.kernel_size)
    stats = conv1x1(sd, "enc_p.proj", h) * mask
    m, logs = torch.split(stats, hp.inter_channels, dim=1)
    return h, m, logs, mask


# --------------------------------------------------------------------------------------------
# modules.py: DDSConv, spline, ConvFlow; models.py: SDP, DP

def ddsconv(sd, p, x, mask, n_layers, ksize, g=None):
    """reference modules.py:118-130 (erf GELU, dropout inactive)."""
    if g is not None:
        x = x + g
    C = x.shape[1]
    for i in range(n_layers):
        dil = ksize ** i
        pad = (ksize * dil - dil) // 2
        y = F.conv1d(x * mask, sd[f"{p}.convs_sep.{i}.weight"], sd[f"{p}.convs_sep.{i}.bias"],
                     padding=pad, dilation=dil, groups=C)
        y = F.gelu(channel_layer_norm(y, sd[f"{p}.norms_1.{i}.gamma"], sd[f"{p}.norms_1.{i}.beta"]))
        y = conv1x1(sd, f"{p}.convs_1x1.{i}", y)
        y = F.gelu(channel_layer_norm(y, sd[f"{p}.norms_2.{i}.gamma"], sd[f"{p}.norms_2.{i}.beta"]))
        x = x + y
    return x * mask


def rq_spline_inverse(y, uw, uh, ud, tail_bound=5.0, min_w=1e-3, min_h=1e-3, min_d=1e-3):
    """Inverse piecewise rational-quadratic spline with linear tails, reference transforms.py:49-96 and
    :99-187 (inverse branch :160-173).  y [...], uw/uh [...,K], ud [...,K-1] → x [...]."""
    K = uw.shape[-1]
    inside = (y >= -tail_bound) & (y <= tail_bound)
    c = math.log(math.exp(1 - min_d) - 1)
    ud = F.pad(ud, (1, 1), value=c)                                    # transforms.py:70-73
    left, right = -tail_bound, tail_bound
    w = min_w + (1 - min_w * K) * torch.softmax(uw, -1)
    cw = F.pad(torch.cumsum(w, -1), (1, 0)) * (right - left) + left
    cw[..., 0], cw[..., -1] = left, right
    w = cw[..., 1:] - cw[..., :-1]
    dv = min_d + F.softplus(ud)
    h = min_h + (1 - min_h * K) * torch.softmax(uh, -1)
    ch = F.pad(torch.cumsum(h, -1), (1, 0)) * (right - left) + left
    ch[..., 0], ch[..., -1] = left, right
    h = ch[..., 1:] - ch[..., :-1]
    knots = ch.clone()
    knots[..., -1] += 1e-6                                             # transforms.py:44-46
    yc = torch.where(inside, y, torch.zeros_like(y))                   # outside values take the identity branch
    b = ((yc[..., None] >= knots).sum(-1) - 1).clamp(0, K - 1)[..., None]
    g = lambda t: t.gather(-1, b)[..., 0]
    icw, iw, ich, ih = g(cw), g(w), g(ch), g(h)
    idl = g(h / w)
    d0, d1 = g(dv), g(dv[..., 1:])
    t = (yc - ich)
    s = d0 + d1 - 2 * idl
    a = t * s + ih * (idl - d0)
    bq = ih * d0 - t * s
    cq = -idl * t
    root = (2 * cq) / (-bq - torch.sqrt(bq * bq - 4 * a * cq))
    out = root * iw + icw
    return torch.where(inside, out, y)


def convflow_reverse(sd, p, z, mask, g, n_layers=3, ksize=3, num_bins=10, tail_bound=5.0):
    """reference modules.py:486-516 with reverse=True. z [B,2,T]."""
    x0, x1 = z[:, :1], z[:, 1:]
    fc = sd[p + ".pre.weight"].shape[0]
    h = conv1x1(sd, p + ".pre", x0)
    h = ddsconv(sd, p + ".convs", h, mask, n_layers, ksize, g=g)
    h = conv1x1(sd, p + ".proj", h) * mask                              # [B,29,T]
    h = h.transpose(1, 2)                                               # [B,T,29]
    uw = h[..., :num_bins] / math.sqrt(fc)
    uh = h[..., num_bins:2 * num_bins] / math.sqrt(fc)
    ud = h[..., 2 * num_bins:]
    x1n = rq_spline_inverse(x1[:, 0], uw, uh, ud, tail_bound)[:, None]
    return torch.cat([x0, x1n], 1) * mask


def sdp_reverse(sd, x, mask, g, noise_w, noise_scale_w):
    """StochasticDurationPredictor.forward(reverse=True), reference models.py:197-204, 245-256.
    Flow order after reversal and dropping the 'useless vflow' (models.py:246-247):
    Flip, CF(flows.7), Flip, CF(flows.5), Flip, CF(flows.3), Flip, ElementwiseAffine(flows.0)."""
    h = conv1x1(sd, "sdp.pre", x) + conv1x1(sd, "sdp.cond", g)
    h = ddsconv(sd, "sdp.convs", h, mask, 3, 3)
    h = conv1x1(sd, "sdp.proj", h) * mask
    z = noise_w * noise_scale_w
    for f in (7, 5, 3):
        z = torch.flip(z, [1])
        z = convflow_reverse(sd, f"sdp.flows.{f}", z, mask, h)
    z = torch.flip(z, [1])
    z = (z - sd["sdp.flows.0.m"]) * torch.exp(-sd["sdp.flows.0.logs"]) * mask   # modules.py:397-399
    return z[:, :1]


def duration_predictor(sd, x, mask, g):
    """reference models.py:285-299."""
    h = x + conv1x1(sd, "dp.cond", g)
    k = sd["dp.conv_1.weight"].shape[2]
    h = torch.relu(F.conv1d(h * mask, sd["dp.conv_1.weight"], sd["dp.conv_1.bias"], padding=k // 2))
    h = channel_layer_norm(h, sd["dp.norm_1.gamma"], sd["dp.norm_1.beta"])
    h = torch.relu(F.conv1d(h * mask, sd["dp.conv_2.weight"], sd["dp.conv_2.bias"], padding=k // 2))
    h = channel_layer_norm(h, sd["dp.norm_2.gamma"], sd["dp.norm_2.beta"])
    return conv1x1(sd, "dp.proj", h * mask) * mask


# --------------------------------------------------------------------------------------------
# length regulation (models.py:1055-1071, commons.py:126-140)

def length_regulate(w_ceil, x_mask, m_p, logs_p):
    """frame j of utterance b belongs to symbol i iff cum[i-1] <= j < cum[i]  (SURVEY.md A.10);
    returns y_lengths, y_mask [B,1,Ty], attn [B,1,Ty,T], expanded m_p/logs_p [B,C,Ty]."""
    B, _, T = w_ceil.shape
    y_lengths = torch.clamp_min(w_ceil.sum([1, 2]), 1).long()
    Ty = int(y_lengths.max())
    y_mask = sequence_mask(y_lengths, Ty)[:, None, :].to(w_ceil.dtype)
    cum = torch.cumsum(w_ceil[:, 0], -1)                                # [B,T]
    j = torch.arange(Ty, dtype=w_ceil.dtype)
    lo = F.pad(cum, (1, 0))[:, :-1]
    attn = ((j[None, :, None] >= lo[:, None, :]) & (j[None, :, None] < cum[:, None, :])).to(w_ceil.dtype)
    attn = attn * y_mask.transpose(1, 2) * x_mask                       # [B,Ty,T]
    idx = attn.argmax(-1)                                               # [B,Ty]
    has = attn.sum(-1, keepdim=False) > 0
    gm = torch.gather(m_p, 2, idx[:, None, :].expand(-1, m_p.shape[1], -1)) * has[:, None, :]
    gl = torch.gather(logs_p, 2, idx[:, None, :].expand(-1, m_p.shape[1], -1)) * has[:, None, :]
    return y_lengths, y_mask, attn[:, None], gm, gl


# --------------------------------------------------------------------------------------------
# flows (models.py:82-145, 403-445; modules.py:133-218, 402-456, 519-580)

def wn(sd, p, x, mask, g, n_layers, hidden, ksize=5, dilation_rate=1, fold_cache=None, half=False):
    """reference modules.py:185-210; gate = tanh(first half)·sigmoid(second half) (commons.py:98-105).

    half=True pins the rounding points of the fp16 product path (bv2_set_flow_dtype(BV2_F16) with the residual flow): the
    inputs and weights of in_layers / res_skip_layers are rounded to fp16 (the gate output is STORED as fp16: it is only ever
    a conv input), accumulation, bias, the conditioning slice, tanh / sigmoid, the residual stream x and the skip sum stay
    fp32; cond_layer (a [B,512] x [1536,512] product) stays fp32.  The reference's counterpart is `flow` under
    torch.autocast(float16) (oracle/ref_import.py reference_autocast_runs), which also keeps fp16 partial sums."""
    q = _h if half else (lambda t: t)
    out = torch.zeros_like(x)
    gc = F.conv1d(g, _fw(sd, p + ".cond_layer", fold_cache), sd[p + ".cond_layer.bias"])
    for i in range(n_layers):
        dil = dilation_rate ** i
        pad = (ksize * dil - dil) // 2
        xin = F.conv1d(q(x), q(_fw(sd, f"{p}.in_layers.{i}", fold_cache)), sd[f"{p}.in_layers.{i}.bias"],
                       padding=pad, dilation=dil)
        a = xin + gc[:, i * 2 * hidden:(i + 1) * 2 * hidden]
        acts = q(torch.tanh(a[:, :hidden]) * torch.sigmoid(a[:, hidden:]))
        rs = F.conv1d(acts, q(_fw(sd, f"{p}.res_skip_layers.{i}", fold_cache)), sd[f"{p}.res_skip_layers.{i}.bias"])
        if i < n_layers - 1:
            x = (x + rs[:, :hidden]) * mask
            out = out + rs[:, hidden:]
        else:
            out = out + rs
    return out * mask


def _fw(sd, prefix, cache):
    if cache is None:
        return fold_weight_norm(sd, prefix)
    if prefix not in cache:
        cache[prefix] = fold_weight_norm(sd, prefix)
    return cache[prefix]


def flow_reverse(sd, hp, z_p, y_mask, g, fold_cache=None, flow_dtype="fp32"):
    """TransformerCouplingBlock / ResidualCouplingBlock reverse: for each coupling, Flip first, then the
    coupling (reference models.py:143-144, 443-444; modules.py:374-381, 561-580, 437-456); mean_only."""
    from_flows = hp.n_flow_layer if hp.use_transformer_flow else 4
    half = hp.inter_channels // 2
    x = z_p
    for f in reversed(range(from_flows)):
        p = f"flow.flows.{2 * f}"
        x = torch.flip(x, [1])
        x0, x1 = x[:, :half], x[:, half:]
        h = conv1x1(sd, p + ".pre", x0) * y_mask
        if hp.use_transformer_flow:
            h = encoder(sd, p + ".enc", h, y_mask, g, hp.n_layers_trans_flow, hp.n_heads, 5, half=(flow_dtype == "fp16"))
        else:
            h = wn(sd, p + ".enc", h, y_mask, g, hp.n_flow_layer, hp.hidden_channels, 5, 1, fold_cache, half=(flow_dtype == "fp16"))
        m = conv1x1(sd, p + ".post", h) * y_mask
        x1 = (x1 - m) * y_mask
        x = torch.cat([x0, x1], 1)
    return x


# --------------------------------------------------------------------------------------------
# Generator (models.py:538-557, modules.py:296-309)

def resblock1(sd, p, x, k, dilations, fold_cache=None):
    for m, d in enumerate(dilations):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _fw(sd, f"{p}.convs1.{m}", fold_cache), sd[f"{p}.convs1.{m}.bias"],
                      padding=(k * d - d) // 2, dilation=d)
        xt = F.leaky_relu(xt, LRELU_SLOPE)
        xt = F.conv1d(xt, _fw(sd, f"{p}.convs2.{m}", fold_cache), sd[f"{p}.convs2.{m}.bias"], padding=(k - 1) // 2)
        x = xt + x
    return x


def resblock2(sd, p, x, k, dilations, fold_cache=None):
    """modules.ResBlock2.forward without a mask (reference modules.py:348-357): one conv per dilation, dilation[0] and dilation[1]."""
    for m, d in enumerate(dilations[:2]):
        xt = F.leaky_relu(x, LRELU_SLOPE)
        xt = F.conv1d(xt, _fw(sd, f"{p}.convs.{m}", fold_cache), sd[f"{p}.convs.{m}.bias"], padding=(k * d - d) // 2, dilation=d)
        x = xt + x
    return x


def _resblock(hp):
    return resblock2 if str(getattr(hp, "resblock", "1")) == "2" else resblock1


def generator(sd, hp, z, g, fold_cache=None, taps: Optional[dict] = None):
    x = F.conv1d(z, sd["dec.conv_pre.weight"], sd["dec.conv_pre.bias"], padding=3) + conv1x1(sd, "dec.cond", g)
    nk = len(hp.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
        x = F.leaky_relu(x, LRELU_SLOPE)
        x = F.conv_transpose1d(x, _fw(sd, f"dec.ups.{i}", fold_cache), sd[f"dec.ups.{i}.bias"],
                               stride=u, padding=(k - u) // 2)
        if taps is not None:
            taps[f"dec.ups.{i}"] = x
        xs = None
        for j in range(nk):
            r = _resblock(hp)(sd, f"dec.resblocks.{i * nk + j}", x, hp.resblock_kernel_sizes[j],
                              hp.resblock_dilation_sizes[j], fold_cache)
            xs = r if xs is None else xs + r
        x = xs / nk
        if taps is not None:
            taps[f"dec.stage.{i}"] = x
    x = F.leaky_relu(x)                                                 # default slope 0.01 (models.py:553)
    x = F.conv1d(x, sd["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


# --------------------------------------------------------------------------------------------
# Generator with bf16 storage / fp32 accumulation (BASELINE config 3).  The reference's counterpart is running `dec`
# under torch.autocast(bfloat16); this restatement pins the rounding points of the bf16 product path so parity can be
# held tightly: weights, (z*mask) and every stored activation are rounded to bf16 (round-to-nearest-even), each conv
# input is bf16(leaky_relu(.)), accumulation + bias + residual are fp32, conv_post/tanh are fp32 on bf16 inputs.  A stage's output — the
# mean of its n branch outputs r_j (models.py:545-552) — is ONE bf16 tensor since round 6 (the stage hand-over, kernels/cl_bf16.h stage_mean):
# the branches are summed widest kernel first with the running sum stored as bf16, mean = bf16((bf16(r_{n-1} + r_{n-2}) + ... + r_0) * fp32(1/n)).

def _bf(x: torch.Tensor) -> torch.Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def resblock1_bf16(sd, p, x, k, dilations, fold_cache=None):
    for m, d in enumerate(dilations):
        xt = _bf(F.leaky_relu(x, LRELU_SLOPE))
        xt = _bf(F.conv1d(xt, _bf(_fw(sd, f"{p}.convs1.{m}", fold_cache)), sd[f"{p}.convs1.{m}.bias"],
                          padding=(k * d - d) // 2, dilation=d))
        xt = _bf(F.leaky_relu(xt, LRELU_SLOPE))
        x = _bf(F.conv1d(xt, _bf(_fw(sd, f"{p}.convs2.{m}", fold_cache)), sd[f"{p}.convs2.{m}.bias"],
                         padding=(k - 1) // 2) + x)
    return x


def resblock2_bf16(sd, p, x, k, dilations, fold_cache=None):
    for m, d in enumerate(dilations[:2]):
        xt = _bf(F.leaky_relu(x, LRELU_SLOPE))
        x = _bf(F.conv1d(xt, _bf(_fw(sd, f"{p}.convs.{m}", fold_cache)), sd[f"{p}.convs.{m}.bias"], padding=(k * d - d) // 2, dilation=d) + x)
    return x


def stage_mean_bf16(rs, inv):
    """The stage hand-over's rounding points (kernels/cl_bf16.h stage_mean / stage_accum): branches summed last (widest kernel) first, the
    running sum stored as bf16 between branches, the mean rounded once more."""
    if len(rs) == 1:
        return rs[0]
    s = rs[-1]
    for r in rs[-2:0:-1]:
        s = _bf(s + r)
    return _bf((s + rs[0]) * inv)


def generator_bf16(sd, hp, z, g, fold_cache=None, taps: Optional[dict] = None):
    x = _bf(F.conv1d(_bf(z), _bf(sd["dec.conv_pre.weight"]), sd["dec.conv_pre.bias"], padding=3) + conv1x1(sd, "dec.cond", g))
    nk = len(hp.resblock_kernel_sizes)
    inv = torch.tensor(1.0 / nk, dtype=torch.float32)
    for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
        x = _bf(F.leaky_relu(x, LRELU_SLOPE))
        x = _bf(F.conv_transpose1d(x, _bf(_fw(sd, f"dec.ups.{i}", fold_cache)), sd[f"dec.ups.{i}.bias"],
                                   stride=u, padding=(k - u) // 2))
        if taps is not None:
            taps[f"dec.ups.{i}"] = x
        rs = []
        for j in range(nk):
            rb = resblock2_bf16 if str(getattr(hp, "resblock", "1")) == "2" else resblock1_bf16
            rs.append(rb(sd, f"dec.resblocks.{i * nk + j}", x, hp.resblock_kernel_sizes[j], hp.resblock_dilation_sizes[j], fold_cache))
        x = stage_mean_bf16(rs, inv)
        if taps is not None:
            taps[f"dec.stage.{i}"] = x
    x = F.leaky_relu(x)                                                 # default slope 0.01 (models.py:553), fp32
    x = F.conv1d(x, sd["dec.conv_post.weight"], None, padding=3)
    return torch.tanh(x)


# --------------------------------------------------------------------------------------------
# the whole path

@torch.no_grad()
def infer(sd, hp, x, x_lengths, sid, tone, language, bert, ja_bert, en_bert, *, noise_w, noise_z,
          noise_scale=0.667, length_scale=1.0, noise_scale_w=0.8, max_len=None, sdp_ratio=0.0,
          w_ceil_override=None, fold_cache=None, want_taps=False, generator_dtype="fp32", flow_dtype="fp32"):
    """reference models.py:1026-1074 with both RNG draws made explicit inputs:
    noise_w [B,2,T] replaces models.py:248-251, noise_z [B,C,>=T_y] replaces randn_like at :1071.
    Returns a dict with the reference's return values plus intermediate taps."""
    g = F.embedding(sid, sd["emb_g.weight"])[:, :, None]
    h, m_p, logs_p, x_mask = text_encoder(sd, hp, x, x_lengths, tone, language, bert, ja_bert, en_bert, g)
    logw_sdp = sdp_reverse(sd, h, x_mask, g, noise_w, noise_scale_w)
    logw_dp = duration_predictor(sd, h, x_mask, g)
    logw = logw_sdp * sdp_ratio + logw_dp * (1 - sdp_ratio)
    w = torch.exp(logw) * x_mask * length_scale
    w_ceil = torch.ceil(w)
    if w_ceil_override is not None:
        w_ceil = w_ceil_override.to(w_ceil.dtype)
    y_lengths, y_mask, attn, m_e, logs_e = length_regulate(w_ceil, x_mask, m_p, logs_p)
    Ty = y_mask.shape[2]
    z_p = m_e + noise_z[:, :, :Ty] * torch.exp(logs_e) * noise_scale
    z = flow_reverse(sd, hp, z_p, y_mask, g, fold_cache, flow_dtype)
    taps = {} if want_taps else None
    gen = generator_bf16 if generator_dtype == "bf16" else generator
    o = gen(sd, hp, (z * y_mask)[:, :, :max_len], g, fold_cache, taps)
    out = dict(o=o, attn=attn, y_mask=y_mask, z=z, z_p=z_p, m_p=m_e, logs_p=logs_e,
               enc_x=h, enc_m=m_p, enc_logs=logs_p, x_mask=x_mask, logw=logw, logw_sdp=logw_sdp, logw_dp=logw_dp,
               w_ceil=w_ceil, y_lengths=y_lengths, g=g)
    if taps:
        out.update(taps)
    return out
