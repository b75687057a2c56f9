"""Model hyper-parameters of the Bert-VITS2 v2.3 synthesizer, as the hot path needs them.

Mirrors the constructor surface of the reference ``SynthesizerTrn``
(reference models.py:816-842) and the fields of ``configs/config.json`` it is fed
from (reference infer.py:95-101).  Unknown kwargs are swallowed the way the
reference's ``**kwargs`` does.
"""
from __future__ import annotations

import dataclasses
import json
import warnings
from typing import List, Sequence

# symbol-table sizes of reference text/symbols.py:167-183 (v2.3): 112 symbols, 12 tones, 3 languages
N_SYMBOLS = 112
N_TONES = 12
N_LANGUAGES = 3
BERT_DIM = 1024           # reference models.py:363-365
ATTN_WINDOW = 4           # reference attentions.py:46 (window_size default)
COND_LAYER_IDX = 2        # reference attentions.py:69-71
SDP_KERNEL = 3            # reference models.py:926-928 (hidden, 192, 3, 0.5, 4)
SDP_N_FLOWS = 4
SDP_DDS_LAYERS = 3        # reference models.py:171-173
SDP_NUM_BINS = 10         # reference modules.py:465
SDP_TAIL_BOUND = 5.0      # reference modules.py:466
DP_FILTER = 256           # reference models.py:929-931
DP_KERNEL = 3
FLOW_KERNEL = 5           # reference models.py:903-924 (both flow variants use 5)
WN_DILATION_RATE = 1
LRELU_SLOPE = 0.1         # reference modules.py:14

# The hyper-parameter envelope bv2_create() accepts (csrc/bv2_model.cpp validate(), the same numbers) — every range below is exercised on the
# GPU against goldens of the real reference at its corners (oracle/cases.ENVELOPE) and against the oracle in its interior
# (cases.random_hparams); a config outside is rejected with a message rather than run untested.  Lists are value sets, pairs are inclusive ranges.
ENVELOPE = dict(
    hidden_channels=[96, 128, 160, 192, 224, 256],
    head_dim=[32, 64, 96, 128],                          # hidden_channels / n_heads
    filter_channels=[128, 192, 256, 320, 384, 448, 512, 576, 640, 704, 768, 832, 896, 960, 1024],     # multiples of 64
    inter_channels=[64, 96, 128, 160, 192, 224, 256],    # multiples of 32
    kernel_size=[1, 3, 5, 7],
    n_layers=(3, 8),                                     # > cond_layer_idx = 2 (attentions.py:69-75); also n_layers_trans_flow
    n_flow_layer=(1, 8),
    gin_channels=[64, 128, 192, 256, 384, 512, 768],     # multiples of 64 up to 768
    n_resblock_kernels=(1, 3),
    resblock_kernel=[3, 5, 7, 9, 11],
    resblock_dilation=(1, 12),
    n_upsamples=(2, 5),
    upsample_rate=[2, 3, 4, 5, 6, 7, 8],
    upsample_taps_per_phase=(1, 4),                      # kernel / rate, with (kernel - rate) even
    upsample_initial_channel=(64, 512),                  # 1024 would need a K-chunked bf16 ConvTranspose tile (LDS); the fp32 form has no such limit
    final_generator_width=[16, 32, 64],                  # upsample_initial_channel >> n_upsamples
)


@dataclasses.dataclass
class HParams:
    n_vocab: int = N_SYMBOLS
    n_tones: int = N_TONES
    n_languages: int = N_LANGUAGES
    spec_channels: int = 1025
    segment_size: int = 32
    inter_channels: int = 192
    hidden_channels: int = 192
    filter_channels: int = 768
    n_heads: int = 2
    n_layers: int = 6
    kernel_size: int = 3
    p_dropout: float = 0.1
    resblock: str = "1"
    resblock_kernel_sizes: Sequence[int] = (3, 7, 11)
    resblock_dilation_sizes: Sequence[Sequence[int]] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    upsample_rates: Sequence[int] = (8, 8, 2, 2, 2)
    upsample_initial_channel: int = 512
    upsample_kernel_sizes: Sequence[int] = (16, 16, 8, 2, 2)
    n_speakers: int = 850
    gin_channels: int = 512
    use_sdp: bool = True
    n_flow_layer: int = 4
    n_layers_trans_flow: int = 4
    flow_share_parameter: bool = False
    use_transformer_flow: bool = True
    sampling_rate: int = 44100
    hop_length: int = 512

    @property
    def total_upsample(self) -> int:
        r = 1
        for u in self.upsample_rates:
            r *= int(u)
        return r

    def validate(self) -> None:
        if str(self.resblock) not in ("1", "2"):
            # reference models.py:508: `modules.ResBlock1 if resblock == "1" else modules.ResBlock2` — every other value means ResBlock2
            warnings.warn(f"resblock={self.resblock!r}: the reference treats every value but '1' as ResBlock2 (models.py:508); doing the same")
        need = 3 if str(self.resblock) == "1" else 2
        if len(self.resblock_dilation_sizes) != len(self.resblock_kernel_sizes):
            # the reference zips the two lists (models.py:524-527) and silently drops the tail of the longer one; a config like that is a typo
            raise ValueError("resblock_kernel_sizes and resblock_dilation_sizes must have the same length")
        if any(len(d) < need for d in self.resblock_dilation_sizes):
            raise ValueError(f"ResBlock{'1' if need == 3 else '2'} reads dilation[0..{need - 1}] (reference modules.py:208-258, 318-346)")
        if self.flow_share_parameter:
            # the reference itself crashes here: attentions.FFT does not exist (models.py:107)
            raise NotImplementedError("flow_share_parameter=True is broken in the reference (models.py:107)")
        if self.n_speakers < 1:
            raise NotImplementedError("n_speakers == 0 needs ReferenceEncoder (models.py:1047-1048); out of scope")
        if self.gin_channels <= 0:
            raise NotImplementedError("gin_channels must be > 0 (reference models.py:871-882)")
        if self.hidden_channels % self.n_heads:
            raise ValueError("hidden_channels must be divisible by n_heads (attentions.py:223)")
        for k in self.resblock_kernel_sizes:
            if k % 2 == 0:
                raise ValueError("resblock kernels must be odd")


_CTOR_FIELDS = {f.name for f in dataclasses.fields(HParams)}


def from_ctor(n_vocab, spec_channels, segment_size, **kwargs) -> HParams:
    """HParams from the reference constructor arguments (unknown kwargs ignored)."""
    kw = {k: v for k, v in kwargs.items() if k in _CTOR_FIELDS}
    hp = HParams(n_vocab=n_vocab, spec_channels=spec_channels, segment_size=segment_size, **kw)
    return hp


def from_config(cfg: dict) -> HParams:
    """HParams from a parsed ``configs/config.json`` (same derivation as reference infer.py:95-101)."""
    data, model = cfg["data"], dict(cfg["model"])
    hp = from_ctor(
        N_SYMBOLS,
        data["filter_length"] // 2 + 1,
        cfg["train"]["segment_size"] // data["hop_length"],
        n_speakers=data["n_speakers"],
        **model,
    )
    hp.sampling_rate = data["sampling_rate"]
    hp.hop_length = data["hop_length"]
    return hp


def from_config_file(path: str) -> HParams:
    with open(path, "r", encoding="utf-8") as f:
        return from_config(json.load(f))


def default_v23(**overrides) -> HParams:
    """The shapes of reference configs/config.json (version 2.3), without needing the file."""
    hp = HParams()
    for k, v in overrides.items():
        setattr(hp, k, v)
    return hp
