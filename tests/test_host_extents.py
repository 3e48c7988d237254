"""Host logic of the parser path, dry-run on the CPU: every pointer / leading dimension the
Python host passes through the C ABI must cover what the kernel behind it touches
(tests/abi_trace.py).  No compute, no GPU."""
import numpy as np
import pytest
import torch

from tests import cases
from tests import abi_trace
from tests.abi_trace import dry_parser, ExtentError


def _batch(case, enc, key):
    from cube_amd.networks.utils import MorphoCollate
    doc = cases.document(case, enc, key)
    X = MorphoCollate(enc).collate_fn(doc.sentences)
    return doc, X


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case,keys", [(cases.TINY, ("single", "equal", "ragged", "g6", "single")),
                                       (cases.FULL, ("ragged", "full_8x32", "full_2x128", "single"))])
def test_decode_batch_extents(monkeypatch, case, keys, precision):
    """Batches of growing and shrinking shapes through one Parser (the grow-only workspace is
    re-used): all ~60 ABI calls per batch stay inside their tensors."""
    cfg, enc, sd = cases.parser_model(case)
    p, lib = dry_parser(cfg, enc, sd, case["ext"], precision, monkeypatch)
    for key in keys:
        doc, X = _batch(case, enc, key)
        X.pop("y_upos")
        bt = p._prepare(X)
        lib.extra = [bt]
        n0 = len(lib.calls)
        r = p.decode_batch_prepared(bt)
        names = [c[0] for c in lib.calls[n0:]]
        assert names.count("mst") == 1 and names.count("label") == 1 and names.count("biaffine") == 1
        assert names.count("bilstm") + names.count("bilstm_i8") == cfg.rnn_layers
        B, L = bt["B"], bt["L"]
        assert tuple(r["packed"].shape) == (5, B, L) and tuple(r["arc_logits"].shape) == (B, L, L + 1)
        assert tuple(r["upos"].shape) == (B, L) and tuple(r["status"].shape) == (B,)


def test_repeated_words_extents(monkeypatch):
    """The batch as the C featuriser hands it over with `dedup_words`: fewer character-CNN words than tokens, and one
    row-to-row copy inside XW0 for the repeated ones."""
    from cube_amd.networks.utils import MorphoCollate
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p, lib = dry_parser(cfg, enc, sd, case["ext"], "bf16x3", monkeypatch)
    doc = cases.document(case, enc, "g6")
    sents = list(doc.sentences[:9])
    for s in sents[1:]:
        s.words[0].word = sents[0].words[0].word
        s.lang_id = sents[0].lang_id
    h = p._host_batch(sents, MorphoCollate(enc))
    staging, h["staging"] = h["staging"], None  # (no GPU here: no event to hand the pinned buffers back with)
    bt = p._upload(h)
    p._staging_release(staging)
    assert bt["n_dup"] >= 8 and bt["N"] + bt["n_dup"] == sum(len(s.words) for s in sents)
    lib.extra = [bt]
    n0 = len(lib.calls)
    r = p.decode_batch_prepared(bt)
    assert ("gather", bt["n_dup"], p._lay.char_half) in lib.calls[n0:]
    assert tuple(r["packed"].shape) == (5, bt["B"], bt["L"])


def test_forward_and_get_labels_extents(monkeypatch):
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p, lib = dry_parser(cfg, enc, sd, case["ext"], "bf16x3", monkeypatch)
    doc, X = _batch(case, enc, "ragged")
    gold = X["y_upos"]
    outs = p.forward(X)  # with y_upos: gold tags feed the upos embedding (parser.py:185-188)
    assert len(outs) == 8
    B, L = X["x_sent"].shape
    assert tuple(outs[0].shape) == (B, L, L + 1) and tuple(outs[2].shape)[:2] == (B, L)
    assert gold.shape == (B, L)
    heads = [np.zeros(len(s.words), dtype=np.int64) for s in doc.sentences]
    lab = p._get_labels(outs[1], heads)
    assert tuple(lab.shape)[:2] == (B, L)


def test_tracer_catches_a_short_buffer(monkeypatch):
    """The checker itself: shrinking one workspace buffer under the host's feet must be caught."""
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p, lib = dry_parser(cfg, enc, sd, case["ext"], "bf16x3", monkeypatch)
    doc, X = _batch(case, enc, "equal")
    X.pop("y_upos")
    bt = p._prepare(X)
    lib.extra = [bt]
    p.decode_batch_prepared(bt)
    real_get = p._buf.get

    def short_get(name, rows, ld, dtype=torch.bfloat16):
        return real_get(name, rows, ld, dtype) if name != "gx" else torch.zeros(8, dtype=dtype)
    monkeypatch.setattr(p._buf, "get", short_get)
    with pytest.raises(ExtentError):
        p.decode_batch_prepared(bt)


@pytest.mark.parametrize("case", [cases.TINY_TAGGER, cases.FULL_TAGGER])
def test_tagger_extents(monkeypatch, case):
    from tests.abi_trace import dry_net
    from cube_amd.networks.tagger import Tagger
    from cube_amd.networks.utils import MorphoCollate
    cfg, enc, sd = cases.tagger_model(case)
    t, lib = dry_net(Tagger, cfg, enc, sd, case["ext"], "bf16x3", monkeypatch)
    for key in ("single", "ragged", "g6"):
        doc = cases.document(case, enc, key)
        X = MorphoCollate(enc).collate_fn(doc.sentences)
        outs = t.forward(X)  # gold-tag path + aux heads
        B, L = X["x_sent"].shape
        assert len(outs) == 6 and all(tuple(o.shape)[:2] == (B, L) for o in outs)
        X.pop("y_upos")
        r = t.tag_batch(X)
        assert tuple(r["packed"].shape) == (3, B, L)
    names = [c[0] for c in lib.calls]
    assert "bilstm" not in names and "mst" not in names


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", [cases.LEMMA_SMALL, cases.LEMMA], ids=["small", "full"])
def test_lemmatizer_host_extents(case, precision, monkeypatch):
    """Every pointer of the lemmatizer's encoder + a few greedy steps covers what its kernel touches."""
    from cube_amd.networks.lemmatizer import LemmaDataset, Word2TargetCollate
    cfg, enc, sd = cases.lemmatizer_model(case)
    m, lib = abi_trace.dry_lemmatizer(cfg, enc, sd, precision, monkeypatch)
    ds = LemmaDataset(cases.lemma_document(enc, "ragged"), for_training=False)
    col = Word2TargetCollate(enc)
    for fused in (True, False):   # cell updates in the GEMM epilogue (two state buffers) / cube_lstm_cell
        m.fused_cell = fused
        for n in (1, 5, len(ds)):
            X = col.collate_fn([ds[i] for i in range(n)], targets=False)
            oc, ok, steps, logits = m._decode(m._encode(X), max_steps=11, keep_logits=True)
            assert oc.shape == (n, 11) and steps == 11 and logits.shape[0] == 11
        kinds = {c[0] for c in lib.calls}
        assert ("lstm_cell" in kinds) == (not fused)
        assert any(c[0] == "gemm" and c[1] == abi_trace.hip.EPI_LSTM for c in lib.calls)
    assert {"gemm", "gather", "lemma_attend", "lstm_cell"} <= kinds


@pytest.mark.parametrize("precision", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", [cases.TOKENIZER_SMALL, cases.TOKENIZER], ids=["small", "full"])
def test_tokenizer_host_extents(case, precision, monkeypatch):
    """Every pointer of Tokenizer.forward covers what its kernel touches (ragged, equal, one piece, long)."""
    from cube_amd.networks.tokenizer import Tokenizer
    cfg, enc, sd = cases.tokenizer_model(case)
    m, lib = abi_trace.dry_net(Tokenizer, cfg, enc, sd, case["ext"], precision, monkeypatch)
    for key in ("ragged", "equal", "one", "long"):
        X = cases.token_batch(case, enc, key)
        y = m.forward(X)
        assert tuple(y.shape) == (len(X["x_sent_len"]), X["x_input"][0].shape[1], 5)
    assert {"gemm", "gather", "masked_mean", "ext_finish"} <= {c[0] for c in lib.calls}
