"""CPU: the tokenizer oracle (oracle/tokenizer_oracle.py) against the reference's own
`Tokenizer.forward` outputs (tests/golden/g10_tokenizer.npz, make_golden.py g10), and the label
decoding of `process` on hand-made cases."""
import os

import numpy as np
import pytest

from oracle import tokenizer_oracle as to
from tests import cases

CASES = {"small": cases.TOKENIZER_SMALL, "full": cases.TOKENIZER}


@pytest.mark.parametrize("tag", ["small", "full"])
def test_g10_forward_matches_reference(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "g10_tokenizer.npz"))
    cfg, enc, sd = cases.tokenizer_model(CASES[tag])
    net = to.OracleTokenizer(cfg, enc, sd, list(CASES[tag]["ext"]))
    for key in ("ragged", "equal", "one", "long"):
        y = net.forward(cases.token_batch(CASES[tag], enc, key)).numpy()
        np.testing.assert_allclose(y, g["%s/%s" % (tag, key)], rtol=2e-4, atol=2e-5, err_msg=key)


def test_decode_labels_follows_the_reference_rules():
    """tokenizer.py:258-287: 2 ends a token, 3 a multi-word token, 4 a token and the sentence; a token
    of only spaces is dropped; left-over pieces form a last token / sentence."""
    toks = ["▁He", "llo", "▁wor", "ld", "!", "▁Don", "'t", "▁", "go"]
    preds = [1, 2, 1, 2, 4, 1, 3, 2, 1]
    sents, mwes = to.decode_labels(toks, preds)
    assert sents == [["Hello", "world", "!"], ["Don't", "go"]]
    assert mwes == [[False, False, False], [True, False]]
    assert to.decode_labels([], []) == ([], [])
    assert to.decode_labels(["a", "b"], [4, 4]) == ([["a"], ["b"]], [[False], [False]])
