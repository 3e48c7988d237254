"""CPU: the lemmatizer oracle (oracle/lemmatizer_oracle.py) against the golden vectors the
reference itself produced (tests/golden/g9_*, make_golden.py g9)."""
import os

import numpy as np
import pytest
import torch

from oracle import lemmatizer_oracle as lo
from tests import cases

CASES = {"small": cases.LEMMA_SMALL, "full": cases.LEMMA}


@pytest.fixture(scope="module")
def g9(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_lemmatizer.npz"))


def check_sequences(yc, yk, ref_c, ref_k, strict, min_same=0.98):
    """Greedy decoding feeds every arg-max back, so one near-tie flips the rest of that word.  The
    full-size model has clear margins and must agree everywhere (strict); the small one may differ
    in a few words, and only from a step where the top-2 margin of the logits is below 1e-3 (early
    steps) or after >= 64 steps of accumulated rounding drift."""
    got_c, got_k = torch.argmax(yc, -1).numpy(), torch.argmax(yk, -1).numpy()
    if strict:
        assert np.array_equal(got_c, ref_c) and np.array_equal(got_k, ref_k)
        return
    same = (got_c == ref_c)
    bad = np.where(~same.all(1))[0]
    assert len(bad) <= (1 - min_same) * len(same), bad
    for w in bad:
        t = int(np.argmin(same[w]))
        top = torch.topk(yc[w, t].double(), 2).values
        assert t >= 64 or float(top[0] - top[1]) < 1e-3, (w, t, float(top[0] - top[1]))
    ok = same.all(1)
    assert np.array_equal(got_k[ok], ref_k[ok])


@pytest.mark.parametrize("tag,dk,n", [("small", "ragged", 37), ("small", "g6", 128), ("full", "ragged", 37)])
def test_g9_collate_and_greedy_decode_match_reference(g9, tag, dk, n):
    cfg, enc, sd = cases.lemmatizer_model(CASES[tag])
    ex = lo.dataset(cases.lemma_document(enc, dk))[:n]
    X = lo.collate(ex, enc)
    for k in ("x_char", "x_case", "x_lang", "x_upos"):
        assert np.array_equal(X[k].numpy(), g9["%s/%s/%s" % (tag, dk, k)]), k
    yc, yk = lo.OracleLemmatizer(cfg, enc, sd).forward(X)
    ref_c, ref_k = g9["%s/%s/char" % (tag, dk)], g9["%s/%s/case" % (tag, dk)]
    assert yc.shape[:2] == ref_c.shape          # same number of decoding steps (EOL / 20*S failsafe)
    check_sequences(yc, yk, ref_c, ref_k, strict=(tag == "full"))
    np.testing.assert_allclose(yc[:, :16].numpy(), g9["%s/%s/char_logits16" % (tag, dk)], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(yk[:, :16].numpy(), g9["%s/%s/case_logits16" % (tag, dk)], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tag,bs", [("small", 4), ("small", 32), ("full", 32)])
def test_g9_process_lemmas_match_reference(golden_dir, tag, bs):
    cfg, enc, sd = cases.lemmatizer_model(CASES[tag])
    doc = cases.lemma_document(enc, "g6", max_word_len=6)
    lo.OracleLemmatizer(cfg, enc, sd).process(doc, batch_size=bs)
    got = "".join(w.word + "\t" + w.lemma + "\n" for s in doc.sentences for w in s.words)
    with open(os.path.join(golden_dir, "g9_lemmas_%s_bs%d.txt" % (tag, bs)), encoding="utf-8") as f:
        ref = f.read()
    if tag == "full":
        assert got == ref
    else:  # near-ties of the small model (see check_sequences)
        a, b = got.split("\n"), ref.split("\n")
        assert len(a) == len(b)
        assert sum(x != y for x, y in zip(a, b)) <= 0.03 * len(a)


def test_dataset_skips_long_words_like_the_reference():
    """utils.py:109-110: words over 25 characters produce no example (and `process` then runs out of
    lemmas, lemmatizer.py:225-229 — an IndexError in the reference)."""
    cfg, enc, sd = cases.lemmatizer_model(cases.LEMMA_SMALL)
    doc = cases.lemma_document(enc, "single")
    doc.sentences[0].words[0].word = "x" * 26
    assert lo.dataset(doc) == []


def test_host_collate_and_strings_match_oracle():
    """cube_amd's vectorised Word2TargetCollate / lemma string assembly == the literal restatements."""
    from cube_amd.networks.lemmatizer import LemmaDataset, Lemmatizer, Word2TargetCollate
    cfg, enc, sd = cases.lemmatizer_model(cases.LEMMA_SMALL)
    doc = cases.lemma_document(enc, "g6")
    for s in doc.sentences:
        for w in s.words:
            w.lemma = w.word[::-1].lower()
    ds = LemmaDataset(doc, for_training=False)
    items = [ds[i] for i in range(len(ds))]
    assert [e["word"] for e in items] == [e["word"] for e in lo.dataset(doc)]
    X = Word2TargetCollate(enc).collate_fn(items)
    ref = lo.collate(items, enc)
    for k in ref:
        assert X[k].dtype == torch.int64 and np.array_equal(X[k].numpy(), ref[k].numpy()), k
    # targets (utils.py:200-213): characters of the lemma, then EOL, zero padded
    C = len(enc.char2int)
    T = max(len(e["target"]) for e in items) + 1
    assert tuple(X["y_char"].shape) == (len(items), T)
    for i, e in enumerate(items):
        want = [enc.char2int.get(ch.lower(), 1) for ch in e["target"]] + [C]
        assert X["y_char"][i, :len(want)].tolist() == want and not X["y_char"][i, len(want):].any()
        assert X["y_case"][i, :len(e["target"])].tolist() == [lo._case(ch) for ch in e["target"]]
    # strings: random ids including <PAD>/<UNK> (multi-character entries) and every case id
    rng = np.random.default_rng(0)
    oc = rng.integers(0, C + 1, size=(500, 12)).astype(np.int32)
    ok = rng.integers(0, 4, size=(500, 12)).astype(np.int32)
    got = Lemmatizer(cfg, enc)._strings(oc, ok)
    onehot_c = torch.nn.functional.one_hot(torch.from_numpy(oc).long(), C + 2).float()
    onehot_k = torch.nn.functional.one_hot(torch.from_numpy(ok).long(), 4).float()
    assert got == lo.OracleLemmatizer(cfg, enc, sd).lemmas(onehot_c, onehot_k)
    with pytest.raises(IndexError):
        Lemmatizer(cfg, enc)._strings(np.full((1, 3), C + 1, dtype=np.int32), np.zeros((1, 3), dtype=np.int32))
