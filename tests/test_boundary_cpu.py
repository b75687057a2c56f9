"""CPU: the drop-in boundary —
 * the RNG contract: the shim's two draws reproduce what a seeded run of the REAL reference drew (fixture
   tests/golden/seeded_*.npz, written by oracle/gen_golden.py from an un-injected ``torch.manual_seed(s); net.infer(...)``);
 * the reference's OWN callers run unchanged on the shim: ``utils.save_checkpoint`` -> ``utils.load_checkpoint`` and the body of
   ``infer.get_net_g`` (infer.py:84-104) — live reference needed, skipped where /root/reference is absent (the GPU box);
 * the reference's ONNX consumer (onnx_modules/V230_OnnxInference/__init__.py) drives ``onnx_api.StageRunner`` objects through its
   own glue code — the runner protocol / tensor names are what that code expects (runners backed by the oracle here: no GPU);
 * handle hygiene: a blob packed under another pack layout is rejected, repack() refuses never-loaded parameters."""
import ctypes as C
import os
import sys
import types

import numpy as np
import pytest
import torch

from bert_vits2_amd import hparams as H, lib as L, models, onnx_api
from oracle import bv2_oracle as O, cases, ref_import
from tests.helpers import GOLDEN, cached_state_dict, load_golden, rms

needs_ref = pytest.mark.skipif(not ref_import.available(), reason="live reference (/root/reference) not present")


def _seeded():
    z = np.load(os.path.join(GOLDEN, "seeded_" + cases.SEEDED_CASE + ".npz"), allow_pickle=False)
    return {k: torch.from_numpy(z[k]) for k in z.files if k != "meta"}


def test_rng_contract_draws_match_a_seeded_reference_run():
    g = _seeded()
    B, _, T = g["noise_w"].shape
    _, Cc, Ty = g["noise_z"].shape
    assert tuple(g["noise_z_strides"].tolist()) == (Cc * Ty, 1, Cc)        # the reference draws on a transposed view
    torch.manual_seed(cases.SEEDED_SEED)
    nw = models.draw_noise_w(B, T, "cpu")
    # between the two draws the reference consumes no randomness (models.py:1052-1071), so the stream position is the same
    nz = models.draw_noise_z(B, Cc, Ty, "cpu")
    assert torch.equal(nw, g["noise_w"])
    assert torch.equal(nz, g["noise_z"]) and tuple(nz.stride()) == (Cc * Ty, 1, Cc)
    # and the naive recipe does NOT reproduce it: the stride handling is load-bearing
    torch.manual_seed(cases.SEEDED_SEED)
    models.draw_noise_w(B, T, "cpu")
    assert not torch.equal(torch.randn(B, Cc, Ty), g["noise_z"])


def test_seeded_reference_run_is_what_the_oracle_computes_from_those_draws():
    """Closes the loop on CPU: the seeded fixture's outputs are the oracle's outputs for the recorded noise."""
    g = _seeded()
    hp, seed, batch, _nw, _nz, kw = cases.build_case(cases.SEEDED_CASE)
    sd = cached_state_dict(hp, seed)
    out = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"],
                  batch["ja_bert"], batch["en_bert"], noise_w=g["noise_w"], noise_z=g["noise_z"], **kw)
    assert torch.equal(out["attn"], g["attn"]) and torch.equal(out["y_mask"], g["y_mask"])
    assert rms(out["o"] - g["o"]) <= 2e-5


@needs_ref
def test_reference_utils_load_checkpoint_and_get_net_g_body_run_unchanged_on_the_shim(tmp_path):
    ref_import.reference_models()                                         # installs the two sys.modules stubs, REF on sys.path
    import utils as ref_utils                                             # /root/reference/utils.py, unmodified
    hp = H.default_v23()
    sd = cached_state_dict(hp, 0)
    net = ref_import.build_reference_net(hp, sd)
    opt = torch.optim.AdamW(net.parameters(), lr=2e-4)
    path = str(tmp_path / "G_1234.pth")
    ref_utils.save_checkpoint(net, opt, 2e-4, 1234, path)                 # reference utils.py:123-139
    # --- the body of infer.get_net_g (infer.py:84-104) with models.SynthesizerTrn swapped for the shim
    hps = ref_utils.get_hparams_from_file(os.path.join(ref_import.REF, "configs", "config.json"))
    symbols = sys.modules["text"].symbols
    net_g = models.SynthesizerTrn(len(symbols), hps.data.filter_length // 2 + 1, hps.train.segment_size // hps.data.hop_length,
                                  n_speakers=hps.data.n_speakers, **hps.model).to("cpu")
    _ = net_g.eval()
    out = ref_utils.load_checkpoint(path, net_g, None, skip_optimizer=True)
    assert out[0] is net_g and out[3] == 1234
    # every inference tensor arrived; what gets packed is bit-identical to packing the state_dict directly
    got = net_g.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k], v), k
    direct = models.from_hparams(hp)
    direct.load_state_dict(sd, strict=False)
    assert torch.equal(net_g.pack_host_blob(), direct.pack_host_blob())
    # the probing the reference does on the returned object
    assert not hasattr(net_g, "module") and net_g.training is False


class _OracleBackedRunner(onnx_api.StageRunner):
    """StageRunner whose ``_call`` is answered by the oracle on the CPU (the product's own ``_call`` needs a GPU): what is under
    test is the run()/name/shape protocol the reference's consumer code relies on."""

    def __init__(self, sd, hp, stage):
        self.sd, self.hp, self.stage = sd, hp, stage
        self.model = types.SimpleNamespace(device=torch.device("cpu"))

    def _call(self, f):
        sd, hp, s = self.sd, self.hp, self.stage
        if s == "emb_g":
            return [torch.nn.functional.embedding(f["sid"], sd["emb_g.weight"])]
        if s == "enc":
            T = f["x"].shape[1]
            b = [f[k].transpose(0, 1).unsqueeze(0) if f[k].dim() == 2 else f[k] for k in ("bert_0", "bert_1", "bert_2")]
            return list(O.text_encoder(sd, hp, f["x"], torch.tensor([T]), f["t"], f["language"], b[0], b[1], b[2], f["g"]))
        if s == "sdp":
            return [O.sdp_reverse(sd, f["x"], f["x_mask"], f["g"], f["zin"], 1.0)]
        if s == "dp":
            return [O.duration_predictor(sd, f["x"], f["x_mask"], f["g"])]
        if s == "flow":
            return [O.flow_reverse(sd, hp, f["z_p"], f["y_mask"], f["g"])]
        return [O.generator(sd, hp, f["z_in"], f["g"])]


@needs_ref
def test_reference_onnx_consumer_drives_the_stage_runners():
    sys.modules.setdefault("onnxruntime", types.ModuleType("onnxruntime"))   # imported at module level by the consumer; unused here
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "ref_onnx_infer", os.path.join(ref_import.REF, "onnx_modules", "V230_OnnxInference", "__init__.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hp = H.default_v23()
    sd = cached_state_dict(hp, 0)
    sess = object.__new__(mod.OnnxInferenceSession)                            # skip __init__ (it opens .onnx files)
    for s in onnx_api.STAGES:
        setattr(sess, s, _OracleBackedRunner(sd, hp, s))
    T = len(cases.ONNX_INFER_SYMBOLS)
    rng = np.random.RandomState(5)
    seq = np.array(cases.ONNX_INFER_SYMBOLS)
    berts = [rng.randn(T, 1024).astype(np.float32) for _ in range(3)]          # the exported graph's [T, 1024] layout
    kw = dict(seed=7, seq_noise_scale=0.6, sdp_noise_scale=0.9, sdp_ratio=0.5)
    wav_ref_glue = sess(seq, np.zeros(T, np.int64), np.zeros(T, np.int64), *berts, np.array([3]), **kw)
    # the same runners under THIS package's consumer (own glue: gather-form length regulation) give the same audio
    mine = onnx_api.StageSession.__new__(onnx_api.StageSession)
    for s in onnx_api.STAGES:
        setattr(mine, s, _OracleBackedRunner(sd, hp, s))
    wav_own_glue = mine(seq, np.zeros(T, np.int64), np.zeros(T, np.int64), *berts, np.array([3]), **kw)
    assert wav_ref_glue.shape == wav_own_glue.shape and wav_ref_glue.shape[1] == 1
    assert rms(torch.from_numpy(wav_ref_glue - wav_own_glue)) <= 1e-6
    assert rms(torch.from_numpy(wav_ref_glue)) > 0.05
    for s in onnx_api.STAGES:                                                 # tensor names of the exported graphs
        assert tuple(onnx_api.StageRunner.get_inputs(getattr(mine, s))) == onnx_api.INPUT_NAMES[s]


def test_stage_runner_rejects_missing_inputs_and_unknown_stage():
    with pytest.raises(ValueError):
        onnx_api.StageRunner(None, "vocoder")
    r = onnx_api.StageRunner(types.SimpleNamespace(device=torch.device("cpu")), "sdp")
    with pytest.raises(ValueError, match="zin"):
        r.run(None, {"x": np.zeros((1, 192, 4), np.float32), "x_mask": np.ones((1, 1, 4), np.float32), "g": np.zeros((1, 512, 1), np.float32)})


def test_pack_layout_is_stamped_in_the_blob_header_and_never_loaded_params_are_refused():
    hp = H.default_v23()
    m = models.from_hparams(hp)
    with pytest.raises(RuntimeError, match="GPU|no weights"):
        m.repack()
    m.load_state_dict(cached_state_dict(hp, 0), strict=False)
    blob = m.pack_host_blob()
    hdr = blob[:32].numpy().view(np.uint32)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "bv2.h")).read()
    import re
    layout = int(re.search(r"#define BV2_PACK_LAYOUT (\d+)", src).group(1))
    assert hdr[1] == L.ABI_VERSION and hdr[3] == layout
    # placeholders before a load: gamma / weight_g at 1 (the reference's defaults), not 0
    fresh = models.from_hparams(hp)
    assert float(fresh.state_dict()["enc_p.encoder.norm_layers_1.0.gamma"].min()) == 1.0
    assert float(fresh.state_dict()["dec.ups.0.weight_g"].min()) == 1.0


def test_checkpoint_loader_uses_the_safe_unpickler(tmp_path):
    """A checkpoint that smuggles a non-tensor object is refused unless the caller opts in."""
    from bert_vits2_amd import checkpoint

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))

    hp = H.default_v23()
    sd = cached_state_dict(hp, 0)
    p = tmp_path / "evil.pth"
    torch.save(dict(model=dict(sd), iteration=1, optimizer=None, learning_rate=1e-4, extra=Evil()), p)
    m = models.from_hparams(hp)
    with pytest.raises(Exception):
        checkpoint.load_checkpoint(str(p), m, None, skip_optimizer=True)
    checkpoint.load_checkpoint(str(p), m, None, skip_optimizer=True, trust_pickle=True)
