"""CPU: host logic of the batched serving glue (length bucketing, collation, argument checks)."""
import pytest
import torch

from bert_vits2_amd import hparams as H, serving


def _utt(T, sid=0, seed=0):
    g = torch.Generator().manual_seed(seed + T)
    f = lambda: torch.randn(H.BERT_DIM, T, generator=g)
    return serving.Utterance(torch.randint(1, 100, (T,), generator=g), torch.randint(0, 6, (T,), generator=g),
                             torch.zeros(T, dtype=torch.int64), f(), f(), f(), sid)


def test_plan_batches_partitions_and_bounds_padding():
    lengths = [5, 120, 64, 66, 7, 300, 65, 6, 128, 61]
    batches = serving.plan_batches(lengths, max_batch=3, max_pad_ratio=1.25)
    flat = sorted(i for b in batches for i in b)
    assert flat == list(range(len(lengths)))
    for b in batches:
        ls = [lengths[i] for i in b]
        assert len(b) <= 3 and max(ls) <= 1.25 * min(ls)
    assert serving.plan_batches([], 4) == []
    assert serving.plan_batches([10, 10, 10], max_batch=8) == [[0, 1, 2]]
    with pytest.raises(ValueError):
        serving.plan_batches([1], max_batch=0)


def test_collate_pads_with_zeros_and_keeps_lengths():
    us = [_utt(9, 3), _utt(4, 1), _utt(6, 2)]
    b = serving.collate(us, "cpu")
    assert b["x"].shape == (3, 9) and b["bert"].shape == (3, H.BERT_DIM, 9)
    assert b["x_lengths"].tolist() == [9, 4, 6] and b["sid"].tolist() == [3, 1, 2]
    assert torch.equal(b["x"][1, :4], us[1].phones) and int(b["x"][1, 4:].abs().sum()) == 0
    assert torch.equal(b["en_bert"][2, :, :6], us[2].en_bert) and float(b["en_bert"][2, :, 6:].abs().sum()) == 0.0


def test_utterance_validates_shapes_like_the_reference():
    u = _utt(5)
    with pytest.raises(ValueError):
        serving.Utterance(u.phones, u.tones[:4], u.lang_ids, u.bert, u.ja_bert, u.en_bert)
    with pytest.raises(ValueError):
        serving.Utterance(u.phones, u.tones, u.lang_ids, u.bert[:, :4], u.ja_bert, u.en_bert)   # infer.py:124


def test_synthesize_refuses_cpu_model():
    from bert_vits2_amd import models
    m = models.from_hparams(H.default_v23())
    with pytest.raises(RuntimeError):
        serving.synthesize(m, [_utt(5)])
