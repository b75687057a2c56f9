"""GPU: the device BERT feature extractor (bv2_bert_forward through bert_encoder.BertEncoder) against
 (a) the committed goldens of the REAL transformers.BertModel (tests/golden/bert_*.npz, oracle/gen_bert_golden.py), and
 (b) the oracle restatement (oracle/bert_oracle.py, fp64) on seeded inputs — including one run at the full size of the reference's
     chinese-roberta-wwm-ext-large (24 layers x 1024, hidden_states[-3]) and a padded batch.
Bar: fp32 round-off — max-abs error <= 2e-4 on O(1) LayerNorm outputs (summation order differs: split-K slabs, flash softmax)."""
import os

import numpy as np
import pytest
import torch

from oracle import bert_oracle as BO, deberta_oracle as DO
from oracle.gen_bert_golden import CASES, DEBERTA_CASES

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _encoder(cfg, sd, model_type="bert"):
    from bert_vits2_amd.bert_encoder import BertEncoder
    return BertEncoder(**cfg, model_type=model_type).load_state_dict(sd, device="cuda")


@pytest.mark.parametrize("name", sorted(CASES))
def test_device_bert_matches_the_real_bertmodel_golden(name):
    cfg, lengths, seed, use_tt = CASES[name]
    g = np.load(os.path.join(GOLD, f"bert_{name}.npz"))
    enc = _encoder(cfg, {"bert." + k: v for k, v in BO.synthetic_state_dict(cfg, seed).items()})   # BertForMaskedLM-style keys
    ids, tt, ln = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["token_type_ids"]), torch.from_numpy(g["lengths"])
    am = (torch.arange(ids.shape[1])[None, :] < ln[:, None]).long()
    out = enc(ids.cuda(), token_type_ids=tt.cuda(), attention_mask=am.cuda())
    torch.cuda.synchronize()
    assert out.shape == (ids.shape[0], cfg["hidden_size"], ids.shape[1])
    got = out.cpu().transpose(1, 2)                                # [B, S, C] like hidden_states
    ref = torch.from_numpy(g["hidden_m3"])
    valid = am.bool()[..., None]
    err = ((got - ref).abs() * valid).max().item()
    assert torch.isfinite(got).all() and err < 2e-4, err


def test_full_size_chinese_roberta_large_shape_vs_oracle():
    cfg = BO.LARGE
    sd = BO.synthetic_state_dict(cfg, 3, layers=22)              # only the 22 layers hidden_states[-3] needs
    enc = _encoder(cfg, sd)
    assert enc.layers_run == 22
    ids, ln = BO.synthetic_inputs(cfg, [52], 5)                  # a 50-character sentence + [CLS] / [SEP]
    out = enc(ids.cuda())
    torch.cuda.synchronize()
    ref = BO.hidden_state(sd, cfg, ids, 22, dtype=torch.float64).float()
    err = (out.cpu().transpose(1, 2) - ref).abs().max().item()
    print(f"BERT-large S=52: max-abs error vs fp64 oracle {err:.2e} (output scale {ref.abs().max().item():.2f})")
    assert err < 5e-4, err
    # padded batch of three sentences = the three sentences run alone (valid positions)
    ids3, ln3 = BO.synthetic_inputs(cfg, [52, 17, 33], 6)
    out3 = enc(ids3.cuda(), lengths=ln3.cuda()).cpu()
    for b, n in enumerate(ln3.tolist()):
        alone = enc(ids3[b:b + 1, :n].cuda()).cpu()
        assert (out3[b, :, :n] - alone[0]).abs().max().item() < 2e-4, b


def test_word_level_feature_path_feeds_the_text_encoder_unchanged():
    """BertEncoder output [1024, S] + word2ph index == the reference's repeated [1024, T] matrix (chinese_bert.py:48-60)."""
    from bert_vits2_amd import bert_features as BF
    cfg = BO.MID
    sd = BO.synthetic_state_dict(cfg, 2)
    enc = _encoder(cfg, sd)
    ids, _ = BO.synthetic_inputs(cfg, [12], 9)
    word2ph = [1, 2, 2, 3, 1, 2, 2, 2, 4, 2, 2, 1]
    feat, index = BF.word_level_feature_cs(enc(ids.cuda())[0], word2ph)
    ref = BO.hidden_state(sd, cfg, ids, cfg["num_hidden_layers"] - 2)[0]             # [S, C]
    rep = torch.cat([ref[i].repeat(word2ph[i], 1) for i in range(len(word2ph))], 0).T   # the reference's loop
    assert (BF.expand(feat, index).cpu() - rep).abs().max().item() < 2e-4


@pytest.mark.parametrize("lengths", [[1], [2, 1], [129], [300, 257], [80]])
def test_sequence_length_edges_vs_oracle(lengths):
    """S = 1 (a single [CLS]), one key tile + 1, > 8 key tiles (the attention kernel's looped form), S = max_position."""
    cfg = BO.MID
    cfg2 = dict(cfg, max_position_embeddings=max(cfg["max_position_embeddings"], max(lengths)))
    sd2 = BO.synthetic_state_dict(cfg2, 4)
    enc = _encoder(cfg2, sd2)
    ids, ln = BO.synthetic_inputs(cfg2, lengths, 11)
    out = enc(ids.cuda(), lengths=ln.cuda()).cpu().transpose(1, 2)
    ref = BO.hidden_state(sd2, cfg2, ids, cfg2["num_hidden_layers"] - 2, lengths=ln, dtype=torch.float64).float()
    valid = (torch.arange(ids.shape[1])[None, :] < ln[:, None])[..., None]
    assert torch.isfinite(out).all()
    assert ((out - ref).abs() * valid).max().item() < 2e-4


def test_rejects_sequences_longer_than_the_position_table():
    enc = _encoder(BO.TINY, BO.synthetic_state_dict(BO.TINY, 0))
    with pytest.raises(RuntimeError, match="max_position"):
        enc(torch.zeros(1, BO.TINY["max_position_embeddings"] + 1, dtype=torch.int64).cuda())


def test_replica_shares_the_weights_and_runs_on_its_own_stream():
    cfg = BO.MID
    sd = BO.synthetic_state_dict(cfg, 2)
    enc = _encoder(cfg, sd)
    rep = enc.replica()
    assert rep._blob is enc._blob and rep._h.value != enc._h.value
    ids, ln = BO.synthetic_inputs(cfg, [40, 23], 3)
    a = enc(ids.cuda(), lengths=ln.cuda())
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        b = rep(ids.cuda(), lengths=ln.cuda())
    torch.cuda.synchronize()
    assert torch.equal(a, b)


# ---------------------------------------------------------------------------------------------------------------------------------
# DeBERTa-v2 (the reference's Japanese / English extractors): text/japanese_bert.py:34-43, text/english_bert_mock.py:30-41
@pytest.mark.parametrize("name", sorted(DEBERTA_CASES))
def test_device_deberta_matches_the_real_transformers_golden(name):
    cfg_name, lengths, seed, cls = DEBERTA_CASES[name]
    cfg = getattr(DO, cfg_name)
    g = np.load(os.path.join(GOLD, f"deberta_{name}.npz"))
    sd = DO.synthetic_state_dict(cfg, seed)
    if cls != "DebertaV2Model":
        sd = {"deberta." + k: v for k, v in sd.items()}            # DebertaV2ForMaskedLM-style keys (japanese_bert.py loads that class)
    enc = _encoder(cfg, sd, "deberta-v2")
    ids, ln = torch.from_numpy(g["input_ids"]), torch.from_numpy(g["lengths"])
    out = enc(ids.cuda(), lengths=ln.cuda())
    torch.cuda.synchronize()
    got = out.cpu().transpose(1, 2)
    ref = torch.from_numpy(g["hidden_m3"])
    valid = (torch.arange(ids.shape[1])[None, :] < ln[:, None])[..., None]
    err = ((got - ref).abs() * valid).max().item()
    assert torch.isfinite(got).all() and err < 3e-4, err


@pytest.mark.parametrize("cfg_name,lengths", [("TINY_V3", [1]), ("TINY_JA", [2, 1]), ("TINY_JA", [33]), ("MID_V3", [160]),
                                              ("MID_V3", [97, 129, 40])])
def test_deberta_sequence_length_edges_vs_oracle(cfg_name, lengths):
    cfg = getattr(DO, cfg_name)
    sd = DO.synthetic_state_dict(cfg, 7)
    enc = _encoder(cfg, sd, "deberta-v2")
    g = torch.Generator().manual_seed(3)
    ln = torch.tensor(lengths)
    ids = torch.randint(1, cfg["vocab_size"], (len(lengths), max(lengths)), generator=g)
    ids = ids * (torch.arange(max(lengths))[None, :] < ln[:, None])
    out = enc(ids.cuda(), lengths=ln.cuda()).cpu().transpose(1, 2)
    n = cfg["num_hidden_layers"] - 2
    ref = DO.hidden_state(sd, cfg, ids, n, lengths=ln, dtype=torch.float64).float()
    valid = (torch.arange(ids.shape[1])[None, :] < ln[:, None])[..., None]
    assert torch.isfinite(out).all()
    assert ((out - ref).abs() * valid).max().item() < 3e-4
    # the first layers on their own (embeddings -> layer 0 [-> ConvLayer]) localise a failure
    one = _encoder(dict(cfg, num_hidden_layers=3), sd, "deberta-v2")
    o1 = one(ids.cuda(), lengths=ln.cuda()).cpu().transpose(1, 2)
    r1 = DO.hidden_state(sd, cfg, ids, 1, lengths=ln, dtype=torch.float64).float()
    assert ((o1 - r1).abs() * valid).max().item() < 1e-4


@pytest.mark.parametrize("cfg_name", ["LARGE_JA", "LARGE_V3"])
def test_full_size_deberta_large_vs_oracle(cfg_name):
    """deberta-v2-large-japanese-char-wwm (ConvLayer) and deberta-v3-large at their real size: 24 x 1024, 256 buckets."""
    cfg = dict(getattr(DO, cfg_name), vocab_size=2000)            # the vocabulary size only scales the embedding table
    sd = DO.synthetic_state_dict(cfg, 5, layers=22)
    enc = _encoder(cfg, sd, "deberta-v2")
    assert enc.layers_run == 22
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(1, cfg["vocab_size"], (1, 53), generator=g)
    out = enc(ids.cuda())
    torch.cuda.synchronize()
    ref = DO.hidden_state(sd, cfg, ids, 22, dtype=torch.float64).float()
    err = (out.cpu().transpose(1, 2) - ref).abs().max().item()
    print(f"{cfg_name} S=53: max-abs error vs fp64 oracle {err:.2e} (output scale {ref.abs().max().item():.2f})")
    assert err < 1e-3, err
