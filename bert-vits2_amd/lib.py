"""ctypes binding of ``libbv2.so`` (C ABI in ``include/bv2.h``) — the stub a reference maintainer would add.

There is no fallback: if the library cannot be built/loaded this raises, loudly.  Nothing in here computes.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

from . import build as _build
from . import hparams as H

MAX_UPS = 8
MAX_RBK = 4
MAX_RBD = 4

F32, F16, BF16 = 0, 1, 2
ABI_VERSION = 3      # include/bv2.h BV2_ABI_VERSION


class Config(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32),
        ("n_vocab", C.c_int32), ("n_tones", C.c_int32), ("n_languages", C.c_int32), ("bert_dim", C.c_int32),
        ("inter_channels", C.c_int32), ("hidden_channels", C.c_int32), ("filter_channels", C.c_int32),
        ("n_heads", C.c_int32), ("n_layers", C.c_int32), ("kernel_size", C.c_int32),
        ("gin_channels", C.c_int32), ("n_speakers", C.c_int32),
        ("use_transformer_flow", C.c_int32), ("n_flow_layer", C.c_int32), ("n_layers_trans_flow", C.c_int32),
        ("n_upsamples", C.c_int32),
        ("upsample_rates", C.c_int32 * MAX_UPS),
        ("upsample_kernel_sizes", C.c_int32 * MAX_UPS),
        ("upsample_initial_channel", C.c_int32),
        ("n_resblock_kernels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * MAX_RBK),
        ("n_resblock_dilations", C.c_int32),
        ("resblock_dilation_sizes", (C.c_int32 * MAX_RBD) * MAX_RBK),
        ("resblock_type", C.c_int32),            # appended in round 5: 1 = ResBlock1, 2 = ResBlock2 (the library still takes the shorter struct)
    ]


class BertConfig(C.Structure):
    """include/bv2_bert.h bv2_bert_config"""
    _fields_ = [("struct_bytes", C.c_int32), ("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("num_heads", C.c_int32),
                ("intermediate_size", C.c_int32), ("max_position", C.c_int32), ("type_vocab_size", C.c_int32),
                ("num_layers_run", C.c_int32), ("layer_norm_eps", C.c_float), ("arch", C.c_int32), ("att_span", C.c_int32),
                ("conv_kernel_size", C.c_int32)]


class EncodeIn(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32),
        ("x", C.c_void_p), ("x_lengths", C.c_void_p), ("sid", C.c_void_p), ("tone", C.c_void_p), ("language", C.c_void_p),
        ("bert", C.c_void_p), ("ja_bert", C.c_void_p), ("en_bert", C.c_void_p), ("noise_w", C.c_void_p),
        ("noise_scale_w", C.c_float), ("sdp_ratio", C.c_float), ("length_scale", C.c_float),
        ("bert_index", C.c_void_p * 3), ("bert_cols", C.c_int32 * 3),
    ]


class EncodeOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("g", "x", "m_p", "logs_p", "x_mask", "logw_sdp", "logw_dp", "logw", "w_ceil", "y_lengths")]


class DecodeIn(C.Structure):
    _fields_ = [
        ("B", C.c_int32), ("T", C.c_int32), ("Ty", C.c_int32), ("max_len", C.c_int32),
        ("m_p", C.c_void_p), ("logs_p", C.c_void_p), ("x_mask", C.c_void_p), ("w_ceil", C.c_void_p),
        ("y_lengths", C.c_void_p), ("g", C.c_void_p), ("noise_z", C.c_void_p),
        ("nz_bstride", C.c_int64), ("nz_cstride", C.c_int64), ("nz_tstride", C.c_int64), ("noise_scale", C.c_float),
        ("exact_lengths", C.c_int32),
    ]


class DecodeOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("o", "attn", "y_mask", "z", "z_p", "m_p", "logs_p")]


class ProfileRow(C.Structure):
    _fields_ = [("name", C.c_char * 96), ("launches", C.c_int64), ("total_ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


# every symbol include/bv2.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("bv2_abi_version", C.c_int, []),
    ("bv2_create", C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    ("bv2_destroy", None, [_P]),
    ("bv2_last_error", C.c_char_p, [_P]),
    ("bv2_load_tensor", C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int, C.c_int]),
    ("bv2_packed_bytes", C.c_int64, [_P]),
    ("bv2_pack_weights", C.c_int, [_P, _P, C.c_int64]),
    ("bv2_attach_weights", C.c_int, [_P, _P, C.c_int64]),
    ("bv2_detach_weights", C.c_int, [_P]),
    ("bv2_set_generator_dtype", C.c_int, [_P, C.c_int]),
    ("bv2_set_flow_dtype", C.c_int, [_P, C.c_int]),
    ("bv2_workspace_bytes", C.c_int64, [_P, C.c_int, C.c_int, C.c_int]),
    ("bv2_encode_durations", C.c_int, [_P, _P, C.POINTER(EncodeIn), C.POINTER(EncodeOut), _P, C.c_int64]),
    ("bv2_decode", C.c_int, [_P, _P, C.POINTER(DecodeIn), C.POINTER(DecodeOut), _P, C.c_int64]),
    ("bv2_stage_emb_g", C.c_int, [_P, _P, C.c_int, _P, _P]),
    ("bv2_stage_enc_p", C.c_int, [_P, _P, C.c_int, C.c_int] + [_P] * 12 + [_P, C.c_int64]),
    ("bv2_stage_sdp", C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int64]),
    ("bv2_stage_dp", C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int64]),
    ("bv2_stage_flow", C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, _P, C.c_int64]),
    ("bv2_stage_generator", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int64]),
    ("bv2_infer", C.c_int, [_P, _P, C.POINTER(EncodeIn), C.POINTER(EncodeOut), _P, C.c_int64, C.c_int64, C.c_int64, C.c_float,
                            C.c_int32, C.c_int32, C.POINTER(DecodeOut), C.POINTER(C.c_int32), _P, C.c_int64]),
    ("bv2_pcm16", C.c_int, [_P, _P, C.c_int64, _P, C.c_int32, C.c_int32, C.c_int64, _P, C.c_int64, _P]),
    ("bv2_graph_capture_encode", C.c_int, [_P, _P, C.POINTER(EncodeIn), C.POINTER(EncodeOut), _P, C.c_int64, C.POINTER(_P)]),
    ("bv2_graph_capture_decode", C.c_int, [_P, _P, C.POINTER(DecodeIn), C.POINTER(DecodeOut), _P, C.c_int64, C.POINTER(_P)]),
    ("bv2_graph_launch", C.c_int, [_P, _P]),
    ("bv2_graph_num_nodes", C.c_int, [_P]),
    ("bv2_graph_destroy", None, [_P]),
    ("bv2_set_option", C.c_int, [_P, C.c_char_p, C.c_int]),
    ("bv2_set_tap", C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    ("bv2_profile_enable", C.c_int, [_P, C.c_int]),
    ("bv2_profile_reset", C.c_int, [_P]),
    ("bv2_profile_report", C.c_int, [_P, C.POINTER(ProfileRow), C.c_int]),
    # include/bv2_bert.h — BERT feature extractor (reference text/chinese_bert.py:15-37)
    ("bv2_bert_create", C.c_int, [_P, C.POINTER(_P)]),
    ("bv2_bert_destroy", None, [_P]),
    ("bv2_bert_last_error", C.c_char_p, [_P]),
    ("bv2_bert_packed_bytes", C.c_int64, [_P]),
    ("bv2_bert_pack_tensor", C.c_int, [_P, _P, C.c_int64, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    ("bv2_bert_missing", C.c_int, [_P]),
    ("bv2_bert_attach_weights", C.c_int, [_P, _P, C.c_int64]),
    ("bv2_bert_set_option", C.c_int, [_P, C.c_char_p, C.c_int]),
    ("bv2_bert_workspace_bytes", C.c_int64, [_P, C.c_int, C.c_int]),
    ("bv2_bert_forward", C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P, _P, _P, _P, C.c_int64]),
]

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _build.LIB


def load(build_if_missing: bool = True, path: Optional[str] = None) -> C.CDLL:
    """Load libbv2.so (building it with hipcc first if the sources changed).  Raises if impossible.  ``path``: measurement tooling
    only (tools/ab_build.py through ``bench.py --library``) — the FIRST load of the process binds that file (same C ABI) instead of
    the stamped in-tree build; no environment variable changes what the product loads."""
    global _lib
    if _lib is not None:
        if path is not None and os.path.abspath(path) != os.path.abspath(getattr(_lib, "_name", "")):
            raise RuntimeError("libbv2 is already loaded from " + getattr(_lib, "_name", "?"))
        return _lib
    # PyTorch bundles its own libamdhip64.so (soname libamdhip64.so.7).  It must be in the process BEFORE libbv2.so is
    # dlopen'ed so that libbv2's NEEDED libamdhip64.so.7 binds to the SAME HIP runtime that owns torch's device
    # pointers and streams; loaded the other way round the process ends up with two runtimes and the library cannot
    # touch torch's memory (hipMemcpy -> invalid value).
    import torch  # noqa: F401
    if build_if_missing and _build.needs_build():
        try:
            _build.build(verbose=False)
        except RuntimeError as e:
            # no compiler on this box: a library that travelled with the tree is still usable (its ABI version and every
            # symbol are checked below); without one there is nothing to fall back to
            if "hipcc not found" not in str(e) or not os.path.exists(_build.LIB):
                raise
            # ... but only if it was built from THESE sources: a stale library with the right ABI number would run an older
            # packer / kernel set against the current host code
            if os.path.exists(_build.STAMP) and open(_build.STAMP).read().strip() != _build._digest():
                raise RuntimeError("libbv2.so in the tree was built from different sources and there is no hipcc here to "
                                   "rebuild it (python -m bert_vits2_amd.build on a box with ROCm)") from e
            import warnings
            warnings.warn("libbv2.so could not be rebuilt (no hipcc here); loading the library that is in the tree")
    if not os.path.exists(_build.LIB):
        raise RuntimeError(f"{_build.LIB} is missing: the HIP extension must be built (python -m bert_vits2_amd.build); "
                           "there is no CPU fallback")
    if path is None:
        path = _build.LIB
    elif not os.path.exists(path):
        raise RuntimeError(f"{path}: no such library")
    lib = C.CDLL(path)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)         # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.bv2_abi_version() != ABI_VERSION:
        raise RuntimeError("libbv2.so ABI version mismatch")
    _lib = lib
    return lib


def make_config(hp: H.HParams) -> Config:
    hp.validate()
    c = Config()
    c.struct_bytes = C.sizeof(Config)
    c.n_vocab, c.n_tones, c.n_languages, c.bert_dim = hp.n_vocab, hp.n_tones, hp.n_languages, H.BERT_DIM
    c.inter_channels, c.hidden_channels, c.filter_channels = hp.inter_channels, hp.hidden_channels, hp.filter_channels
    c.n_heads, c.n_layers, c.kernel_size = hp.n_heads, hp.n_layers, hp.kernel_size
    c.gin_channels, c.n_speakers = hp.gin_channels, hp.n_speakers
    c.use_transformer_flow = int(bool(hp.use_transformer_flow))
    c.n_flow_layer, c.n_layers_trans_flow = hp.n_flow_layer, hp.n_layers_trans_flow
    ups, uks = list(hp.upsample_rates), list(hp.upsample_kernel_sizes)
    if len(ups) != len(uks) or len(ups) > MAX_UPS:
        raise ValueError("bad upsample configuration")
    c.n_upsamples = len(ups)
    for i, (u, k) in enumerate(zip(ups, uks)):
        c.upsample_rates[i], c.upsample_kernel_sizes[i] = int(u), int(k)
    c.upsample_initial_channel = hp.upsample_initial_channel
    rk, rd = list(hp.resblock_kernel_sizes), [list(d) for d in hp.resblock_dilation_sizes]
    c.resblock_type = 1 if str(hp.resblock) == "1" else 2          # reference models.py:508: anything but "1" is modules.ResBlock2
    # modules.ResBlock1 reads dilation[0..2], ResBlock2 dilation[0..1]; longer lists are ignored by the reference (modules.py:208-258, 318-346),
    # shorter ones raise IndexError there and ValueError in hp.validate().  A direct C caller must hand over exactly 3 / 2 (bv2.h).
    rd = [d[:3] if c.resblock_type == 1 else d[:2] for d in rd]
    if len(rk) > MAX_RBK or len(rd) != len(rk) or any(len(d) != len(rd[0]) or len(d) > MAX_RBD for d in rd):
        raise ValueError("bad resblock configuration")
    c.n_resblock_kernels = len(rk)
    c.n_resblock_dilations = len(rd[0])
    for j, k in enumerate(rk):
        c.resblock_kernel_sizes[j] = int(k)
        for d, v in enumerate(rd[j]):
            c.resblock_dilation_sizes[j][d] = int(v)
    return c
