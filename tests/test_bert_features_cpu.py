"""CPU: the word-level BERT feature pair (bert_vits2_amd/bert_features.py) is the reference's repeat loop
(text/chinese_bert.py:42-60) in index form — expanding it gives exactly the matrix the reference returns, with and without
style-text mixing — and the HF-model wrapper keeps everything on one device."""
import pytest
import torch

from bert_vits2_amd import bert_features as BF


def _reference_repeat(res, word2ph, style_res=None, style_weight=0.7):
    """The loop of reference text/chinese_bert.py:48-60, restated."""
    rows = []
    mean = None if style_res is None else style_res.mean(0)
    for i, n in enumerate(word2ph):
        r = res[i].repeat(n, 1)
        if mean is not None:
            r = res[i].repeat(n, 1) * (1 - style_weight) + mean.repeat(n, 1) * style_weight
        rows.append(r)
    return torch.cat(rows, 0).T


@pytest.mark.parametrize("style", [False, True])
def test_word_level_pair_expands_to_the_reference_matrix(style):
    g = torch.Generator().manual_seed(3)
    word2ph = [1, 2, 1, 2, 2, 1, 4, 2, 3, 1]              # blanks interspersed: word2ph*2 with [0] += 1 (infer.py:117-120)
    S = len(word2ph)
    res = torch.randn(S, 1024, generator=g)
    sres = torch.randn(7, 1024, generator=g) if style else None
    feat, idx = BF.word_level_feature(res, word2ph, sres, 0.7)
    assert feat.shape == (1024, S) and idx.dtype == torch.int32 and idx.shape == (sum(word2ph),)
    want = _reference_repeat(res, word2ph, sres, 0.7)
    assert torch.allclose(BF.expand(feat, idx), want, rtol=0, atol=1e-6)
    if not style:
        assert torch.equal(BF.expand(feat, idx), want)
    assert BF.batch_index([idx, idx[:5]], 24, "cpu").shape == (2, 24)


def test_get_bert_feature_with_a_tiny_hf_model():
    transformers = pytest.importorskip("transformers")
    cfg = transformers.BertConfig(vocab_size=64, hidden_size=1024, num_hidden_layers=3, num_attention_heads=16,
                                  intermediate_size=64, max_position_embeddings=32)
    torch.manual_seed(0)
    model = transformers.BertForMaskedLM(cfg).eval()

    class Tok:                                            # the tokenizer protocol the reference uses: tokenizer(text, return_tensors="pt")
        def __call__(self, text, return_tensors="pt"):
            ids = torch.tensor([[1] + [3 + (ord(ch) % 50) for ch in text] + [2]])
            return {"input_ids": ids, "attention_mask": torch.ones_like(ids), "token_type_ids": torch.zeros_like(ids)}

    text = "abcde"
    word2ph = [1, 2, 2, 3, 2, 2, 1]                       # len(text) + 2 entries (chinese_bert.py:42)
    feat, idx = BF.get_bert_feature(text, word2ph, Tok(), model, "cpu")
    with torch.no_grad():
        res = model(**Tok()(text), output_hidden_states=True)["hidden_states"][-3][0]
    assert torch.equal(BF.expand(feat, idx), _reference_repeat(res, word2ph))
    with pytest.raises(ValueError):
        BF.word_level_feature(res, word2ph[:-1])
