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
