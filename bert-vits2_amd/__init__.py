"""bert-vits2_amd: MI355X-native (gfx950) ``SynthesizerTrn.infer()`` hot path of Bert-VITS2 v2.3.

Only what the hot path needs lives here: ``csrc/`` (hand-written HIP kernels + the C-ABI library
``libbv2.so``), and the host-side mirror of the reference's ``SynthesizerTrn`` interface.
"""
__version__ = "0.1.0"
