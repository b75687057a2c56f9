"""CPU: checkpoint ingest — the loader accepts the reference's checkpoint file format and its
variants, and every variant packs to the same weight blob (the packed blob is what the kernels read)."""
import logging

import pytest
import torch

from bert_vits2_amd import checkpoint, hparams as H, models
from tests.helpers import cached_state_dict


def _fresh(hp):
    return models.from_hparams(hp)


def _blob_from(hp, sd_like, tmp_path, name, **extra):
    path = tmp_path / name
    torch.save(dict(model=sd_like, iteration=7, optimizer=None, learning_rate=2e-4, **extra), path)
    m = _fresh(hp)
    out = checkpoint.load_checkpoint(str(path), m, None, skip_optimizer=True)
    assert out[0] is m and out[2] == 2e-4 and out[3] == 7          # reference return tuple (utils.py:120)
    return m, m.pack_host_blob()


@pytest.fixture(scope="module")
def base():
    hp = H.default_v23()
    sd = cached_state_dict(hp, 3)
    m = _fresh(hp)
    m.load_state_dict(sd, strict=False)
    return hp, sd, m.pack_host_blob()


def test_reference_format_roundtrip_and_training_only_keys(base, tmp_path):
    hp, sd, ref_blob = base
    full = dict(sd)
    full["enc_q.pre.weight"] = torch.zeros(192, 1025, 1)            # training-only tensors are ignored
    full["sdp.post_pre.weight"] = torch.zeros(192, 1, 1)
    m, blob = _blob_from(hp, full, tmp_path, "G_0.pth")
    assert m.last_missing_keys == [] and torch.equal(blob, ref_blob)


def test_ddp_prefix_and_half_release_checkpoint(base, tmp_path):
    hp, sd, ref_blob = base
    _, blob = _blob_from(hp, {"module." + k: v for k, v in sd.items()}, tmp_path, "G_ddp.pth")
    assert torch.equal(blob, ref_blob)
    # compress_model.py:49-53: .half() of every tensor -> packs exactly like the fp16-rounded fp32 weights
    half = {k: v.half() for k, v in sd.items()}
    _, blob16 = _blob_from(hp, half, tmp_path, "G_release.pth")
    m2 = _fresh(hp)
    m2.load_state_dict({k: v.half().float() for k, v in sd.items()}, strict=False)
    assert torch.equal(blob16, m2.pack_host_blob())


def test_folded_weight_norm_checkpoint_packs_to_the_same_weights(base, tmp_path):
    hp, sd, ref_blob = base
    folded = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            continue
        if k.endswith(".weight_v"):
            g = sd[k[:-1] + "g"]
            n = v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1)))
            folded[k[:-2]] = v * (g / n)                              # Generator.remove_weight_norm (models.py:559-564)
        else:
            folded[k] = v
    m, blob = _blob_from(hp, folded, tmp_path, "G_folded.pth")
    assert m.last_missing_keys == []
    got = m.state_dict()
    fold = lambda g, v: v * (g / v.reshape(v.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (v.dim() - 1))))
    n = 0
    for k, v in sd.items():
        if k.endswith(".weight_v"):
            w0, w1 = fold(sd[k[:-1] + "g"], v), fold(got[k[:-1] + "g"], got[k])
            assert (w0 - w1).abs().max() <= 2e-6 * w0.abs().max(), k
            n += 1
    assert n == 95                                                    # 5 ups + 90 resblock convs
    assert blob.numel() == ref_blob.numel()
    same = (blob == ref_blob).float().mean().item()                  # non-weight-normed tensors pack bit-identically
    assert same > 0.5, same


def test_new_parametrization_keys(base, tmp_path):
    hp, sd, ref_blob = base
    ren = {}
    for k, v in sd.items():
        if k.endswith(".weight_g"):
            ren[k[:-len(".weight_g")] + ".parametrizations.weight.original0"] = v
        elif k.endswith(".weight_v"):
            ren[k[:-len(".weight_v")] + ".parametrizations.weight.original1"] = v
        else:
            ren[k] = v
    _, blob = _blob_from(hp, ren, tmp_path, "G_param.pth")
    assert torch.equal(blob, ref_blob)


def test_old_checkpoint_without_ja_bert_proj_is_zero_filled(base, tmp_path, caplog):
    hp, sd, _ = base
    old = {k: v for k, v in sd.items() if "ja_bert_proj" not in k and k != "dp.proj.bias"}
    with caplog.at_level(logging.WARNING):
        m, _ = _blob_from(hp, old, tmp_path, "G_old.pth")
    assert float(m.state_dict()["enc_p.ja_bert_proj.weight"].abs().sum()) == 0.0   # reference utils.py:103-108
    assert m.last_missing_keys == ["dp.proj.bias"]                  # anything else: reported, model value kept
    assert "old version of the model" in caplog.text and "dp.proj.bias is not in the checkpoint" in caplog.text


def test_packed_blob_cache_file(base, tmp_path):
    hp, sd, ref_blob = base
    m = _fresh(hp)
    m.load_state_dict(sd, strict=False)
    n = checkpoint.save_packed(m, str(tmp_path / "G.bv2"))
    assert n == ref_blob.numel() and (tmp_path / "G.bv2").read_bytes() == ref_blob.numpy().tobytes()
    with pytest.raises(RuntimeError):
        checkpoint.load_packed(m, str(tmp_path / "G.bv2"), torch.device("cpu"))   # no CPU fallback
