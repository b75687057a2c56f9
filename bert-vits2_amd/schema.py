"""Reference ``state_dict`` schema of the inference sub-networks.

``param_shapes(hp)`` lists every tensor name/shape that reference
``SynthesizerTrn.infer`` (models.py:1026-1074) reads, in the reference's own key
naming, so that a checkpoint written by reference utils.save_checkpoint
(utils.py:123-141) loads into the shim unchanged.  ``enc_q.*`` (posterior
encoder, training only) is deliberately absent.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

from . import hparams as H


def _conv(d, name, cout, cin, k, bias=True):
    d[name + ".weight"] = (cout, cin, k)
    if bias:
        d[name + ".bias"] = (cout,)


def _wn_conv(d, name, cout, cin, k):
    """old-style torch.nn.utils.weight_norm(dim=0): weight_g [cout,1,1], weight_v [cout,cin,k]."""
    d[name + ".weight_g"] = (cout, 1, 1)
    d[name + ".weight_v"] = (cout, cin, k)
    d[name + ".bias"] = (cout,)


def _ln(d, name, c):
    d[name + ".gamma"] = (c,)
    d[name + ".beta"] = (c,)


def _encoder(d, p, hidden, filt, heads, layers, ksize, gin):
    """attentions.Encoder (reference attentions.py:37-101)."""
    dk = hidden // heads
    d[p + ".spk_emb_linear.weight"] = (hidden, gin)
    d[p + ".spk_emb_linear.bias"] = (hidden,)
    for i in range(layers):
        a = f"{p}.attn_layers.{i}"
        d[a + ".emb_rel_k"] = (1, 2 * H.ATTN_WINDOW + 1, dk)
        d[a + ".emb_rel_v"] = (1, 2 * H.ATTN_WINDOW + 1, dk)
        for c in ("conv_q", "conv_k", "conv_v", "conv_o"):
            _conv(d, f"{a}.{c}", hidden, hidden, 1)
    for i in range(layers):
        _ln(d, f"{p}.norm_layers_1.{i}", hidden)
    for i in range(layers):
        _conv(d, f"{p}.ffn_layers.{i}.conv_1", filt, hidden, ksize)
        _conv(d, f"{p}.ffn_layers.{i}.conv_2", hidden, filt, ksize)
    for i in range(layers):
        _ln(d, f"{p}.norm_layers_2.{i}", hidden)


def _ddsconv(d, p, c, ksize, layers):
    for i in range(layers):
        d[f"{p}.convs_sep.{i}.weight"] = (c, 1, ksize)
        d[f"{p}.convs_sep.{i}.bias"] = (c,)
    for i in range(layers):
        _conv(d, f"{p}.convs_1x1.{i}", c, c, 1)
    for i in range(layers):
        _ln(d, f"{p}.norms_1.{i}", c)
    for i in range(layers):
        _ln(d, f"{p}.norms_2.{i}", c)


def n_coupling_flows(hp: H.HParams) -> int:
    """Transformer flow: n_flows = n_flow_layer (models.py:903-915); residual flow: n_flows stays at its
    default 4 while n_flow_layer becomes the WN depth (models.py:916-923, 403-413)."""
    return hp.n_flow_layer if hp.use_transformer_flow else 4


def param_shapes(hp: H.HParams) -> "OrderedDict[str, Tuple[int, ...]]":
    d: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    hid, inter, filt, gin = hp.hidden_channels, hp.inter_channels, hp.filter_channels, hp.gin_channels
    half = inter // 2

    # ---- enc_p: TextEncoder (reference models.py:333-400)
    d["enc_p.emb.weight"] = (hp.n_vocab, hid)
    d["enc_p.tone_emb.weight"] = (hp.n_tones, hid)
    d["enc_p.language_emb.weight"] = (hp.n_languages, hid)
    for n in ("bert_proj", "ja_bert_proj", "en_bert_proj"):
        _conv(d, f"enc_p.{n}", hid, H.BERT_DIM, 1)
    _encoder(d, "enc_p.encoder", hid, filt, hp.n_heads, hp.n_layers, hp.kernel_size, gin)
    _conv(d, "enc_p.proj", 2 * inter, hid, 1)

    # ---- dec: HiFi-GAN Generator (reference models.py:490-564)
    c0 = hp.upsample_initial_channel
    _conv(d, "dec.conv_pre", c0, inter, 7)
    ch = c0
    for i, (u, k) in enumerate(zip(hp.upsample_rates, hp.upsample_kernel_sizes)):
        cin, cout = c0 // (2 ** i), c0 // (2 ** (i + 1))
        # ConvTranspose1d weight is [cin, cout, k]; weight_norm dim=0 is therefore C_in
        d[f"dec.ups.{i}.weight_g"] = (cin, 1, 1)
        d[f"dec.ups.{i}.weight_v"] = (cin, cout, k)
        d[f"dec.ups.{i}.bias"] = (cout,)
    nk = len(hp.resblock_kernel_sizes)
    for i in range(len(hp.upsample_rates)):
        ch = c0 // (2 ** (i + 1))
        for j, k in enumerate(hp.resblock_kernel_sizes):
            r = i * nk + j
            nd = len(hp.resblock_dilation_sizes[j])
            if str(hp.resblock) == "2":          # modules.ResBlock2 (reference modules.py:318-346): `convs.0`, `convs.1`
                for m in range(2):
                    _wn_conv(d, f"dec.resblocks.{r}.convs.{m}", ch, ch, k)
                continue
            for cs in ("convs1", "convs2"):
                for m in range(nd):
                    _wn_conv(d, f"dec.resblocks.{r}.{cs}.{m}", ch, ch, k)
    _conv(d, "dec.conv_post", 1, ch, 7, bias=False)
    _conv(d, "dec.cond", c0, gin, 1)

    # ---- flow (reference models.py:82-145 / 403-445); coupling layers sit at even indices, Flip at odd
    for f in range(n_coupling_flows(hp)):
        p = f"flow.flows.{2 * f}"
        _conv(d, p + ".pre", hid, half, 1)
        if hp.use_transformer_flow:
            _encoder(d, p + ".enc", hid, filt, hp.n_heads, hp.n_layers_trans_flow, H.FLOW_KERNEL, gin)
        else:
            nl = hp.n_flow_layer  # reference models.py:916-923 passes n_flow_layer as WN n_layers
            _wn_conv(d, p + ".enc.cond_layer", 2 * hid * nl, gin, 1)
            for i in range(nl):
                _wn_conv(d, f"{p}.enc.in_layers.{i}", 2 * hid, hid, H.FLOW_KERNEL)
            for i in range(nl):
                rs = 2 * hid if i < nl - 1 else hid
                _wn_conv(d, f"{p}.enc.res_skip_layers.{i}", rs, hid, 1)
        _conv(d, p + ".post", half, hid, 1)

    # ---- sdp: StochasticDurationPredictor (reference models.py:148-204); post_* (training) omitted
    fc = hid  # reference models.py:159: filter_channels overwritten by in_channels
    d["sdp.flows.0.m"] = (2, 1)
    d["sdp.flows.0.logs"] = (2, 1)
    for f in range(H.SDP_N_FLOWS):
        p = f"sdp.flows.{2 * f + 1}"
        _conv(d, p + ".pre", fc, 1, 1)
        _ddsconv(d, p + ".convs", fc, H.SDP_KERNEL, H.SDP_DDS_LAYERS)
        _conv(d, p + ".proj", 3 * H.SDP_NUM_BINS - 1, fc, 1)
    _conv(d, "sdp.pre", fc, hid, 1)
    _conv(d, "sdp.proj", fc, fc, 1)
    _ddsconv(d, "sdp.convs", fc, H.SDP_KERNEL, H.SDP_DDS_LAYERS)
    _conv(d, "sdp.cond", fc, gin, 1)

    # ---- dp: DurationPredictor (reference models.py:259-299)
    _conv(d, "dp.conv_1", H.DP_FILTER, hid, H.DP_KERNEL)
    _ln(d, "dp.norm_1", H.DP_FILTER)
    _conv(d, "dp.conv_2", H.DP_FILTER, H.DP_FILTER, H.DP_KERNEL)
    _ln(d, "dp.norm_2", H.DP_FILTER)
    _conv(d, "dp.proj", 1, H.DP_FILTER, 1)
    _conv(d, "dp.cond", hid, gin, 1)

    d["emb_g.weight"] = (hp.n_speakers, gin)
    return d


def n_params(hp: H.HParams) -> int:
    n = 0
    for s in param_shapes(hp).values():
        m = 1
        for x in s:
            m *= x
        n += m
    return n
