"""CPU: the pooling oracle (oracle/lm_oracle.py) against golden G12 = the reference's own LMHelperHF.apply driven with the
synthetic tokenizer / transformer stand-ins of tests/cases.py (lm.py:141-208)."""
import os

import numpy as np

from oracle import lm_oracle
from tests import cases


def test_pooling_oracle_equals_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g12_lm_pooling.npz"))
    _, enc, _ = cases.parser_model(cases.FULL)
    doc = cases.lm_document(enc)
    split, model = cases.FakeSplitter(), cases.FakeEncoder()
    n_pieces = [len(split(w.word)["input_ids"]) - 2 for s in doc.sentences for w in s.words]
    assert 0 in n_pieces and 1 in n_pieces and 2 in n_pieces and max(n_pieces) >= 4   # every branch of lm.py:186-194
    assert sum(len(split(w.word)["input_ids"]) - 2 for w in doc.sentences[3].words) + 2 > 512   # two windows
    for i, s in enumerate(doc.sentences):
        we = lm_oracle.compute_we([s], split, model)                       # apply() calls the model per sentence
        a = np.stack(we).reshape(len(s.words), cases.LM_LAYERS, cases.LM_WIDTH)
        exp = g["sent%d" % i]
        np.testing.assert_array_equal(a if i < 3 else a[:, :, ::16], exp)   # bit-equal: same fp32 operations in the same order
    # the quirk itself: the first piece counts twice, the last one never
    w = next(w for s in doc.sentences for w in s.words if len(split(w.word)["input_ids"]) - 2 == 3)
    ids = np.array([[0] + split(w.word)["input_ids"][1:-1] + [2]])
    import torch
    hs = torch.cat(model(torch.from_numpy(ids))["hidden_states"], dim=-1)[0].numpy()

    class S:
        words = [w]
    got = lm_oracle.compute_we([S], split, model)[0]
    np.testing.assert_array_equal(got, ((hs[1] + hs[1]) + hs[2]) / 3)
