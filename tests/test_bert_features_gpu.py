"""GPU: word-level BERT features handed to the TextEncoder front through ``bert_index`` (bv2_encode_in.bert_index)
give the same encoder outputs / audio as the reference's materialised ``[1024, T]`` matrix — the repeat happens as a gather on the
device, the feature matrix stays word-level."""
import pytest
import torch

from bert_vits2_amd import bert_features as BF
from oracle import bv2_oracle as O, cases
from tests.helpers import cached_state_dict, rms

pytestmark = pytest.mark.gpu


def test_word_level_features_through_the_gather_match_the_repeated_matrix():
    from bert_vits2_amd import models
    hp, seed, batch, nw, nz, kw = cases.build_case("mix_b2_ragged")
    sd = cached_state_dict(hp, seed)
    m = models.from_hparams(hp)
    m.load_state_dict(sd, strict=False)
    m = m.to("cuda").eval()
    B, T = batch["x"].shape
    lens = batch["x_lengths"].tolist()
    g = torch.Generator().manual_seed(1)
    # utterance 0 speaks JP (feature 1), utterance 1 EN (feature 2) in this case; make BOTH ja_bert and en_bert word-level
    feats, idxs = {1: [], 2: []}, {1: [], 2: []}
    S = 0
    for f in (1, 2):
        for b in range(B):
            n = lens[b]
            w2p, left = [], n
            while left > 0:
                r = min(left, int(torch.randint(1, 4, (1,), generator=g)))
                w2p.append(r)
                left -= r
            feat, ix = BF.word_level_feature(torch.randn(len(w2p), 1024, generator=g), w2p)
            feats[f].append(feat)
            idxs[f].append(ix)
            S = max(S, feat.shape[1])
    word = {f: torch.zeros(B, 1024, S) for f in (1, 2)}
    full = {f: torch.zeros(B, 1024, T) for f in (1, 2)}
    for f in (1, 2):
        for b in range(B):
            word[f][b, :, : feats[f][b].shape[1]] = feats[f][b]
            full[f][b, :, : lens[b]] = BF.expand(feats[f][b], idxs[f][b])
    index = {f: BF.batch_index(idxs[f], T, "cuda") for f in (1, 2)}
    dev = lambda t: t.cuda()
    common = (dev(batch["x"]), dev(batch["x_lengths"]), dev(batch["sid"]), dev(batch["tone"]), dev(batch["language"]), dev(batch["bert"]))
    a = m.infer(*common, dev(full[1]), dev(full[2]), noise_w=nw, noise_z=nz.cuda(), **kw)
    enc_a = {k: v.clone() for k, v in m.last_encode.items()}
    b_ = m.infer(*common, dev(word[1]), dev(word[2]), noise_w=nw, noise_z=nz.cuda(), bert_index=(None, index[1], index[2]), **kw)
    enc_b = m.last_encode
    torch.cuda.synchronize()
    assert torch.equal(enc_a["w_ceil"], enc_b["w_ceil"])
    for k in ("x", "m_p", "logs_p"):
        assert (enc_a[k] - enc_b[k]).abs().max().item() <= 1e-5 * enc_a[k].abs().max().item(), k
    assert rms((a[0] - b_[0]).cpu()) < 2e-6
    # and against the oracle fed the reference-style repeated matrices
    ref = O.infer(sd, hp, batch["x"], batch["x_lengths"], batch["sid"], batch["tone"], batch["language"], batch["bert"], full[1], full[2],
                  noise_w=nw, noise_z=nz, w_ceil_override=enc_b["w_ceil"].cpu()[:, None], **kw)
    assert rms(b_[0].cpu() - ref["o"]) <= 5e-5
    # graph replay carries the index too
    m.enable_graphs(True, ty_bucket=1)
    c = m.infer(*common, dev(word[1]), dev(word[2]), noise_w=nw, noise_z=nz.cuda(), bert_index=(None, index[1], index[2]), **kw)
    assert torch.equal(c[0], b_[0])
    with pytest.raises(ValueError):
        m.infer(*common, dev(word[1]), dev(word[2]), noise_w=nw, noise_z=nz.cuda(), **kw)       # word-level shape without an index
