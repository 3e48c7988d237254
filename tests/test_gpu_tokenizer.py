"""GPU parity tests of the tokenizer row, through the C ABI: `Tokenizer.forward`
against the reference's own logits (tests/golden/g10_tokenizer.npz, produced by make_golden.py g10
with the reference imported) within the north-star tolerance (1e-3 relative), labels equal wherever
the top-2 margin exceeds it; `process` end to end with a stand-in sub-word collate against the
oracle's label decoding."""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3
CASES = {"small": cases.TOKENIZER_SMALL, "full": cases.TOKENIZER}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from cube_amd import hip
    hip.load()
    return torch.device("cuda:0")


_models = {}


def _model(tag, dev):
    if tag not in _models:
        from cube_amd.networks.tokenizer import Tokenizer
        cfg, enc, sd = cases.tokenizer_model(CASES[tag])
        m = Tokenizer(cfg, enc, ext_word_emb=list(CASES[tag]["ext"]))
        m.load_state_dict(sd)
        m.to(dev)
        _models[tag] = (m, cfg, enc, sd)
    return _models[tag]


@pytest.mark.parametrize("tag", ["small", "full"])
@pytest.mark.parametrize("key", ["ragged", "equal", "one", "long"])
def test_forward_matches_reference_golden(dev, golden_dir, tag, key):
    g = np.load(os.path.join(golden_dir, "g10_tokenizer.npz"))
    m, cfg, enc, sd = _model(tag, dev)
    X = cases.token_batch(CASES[tag], enc, key)
    y = m.forward(X).cpu().numpy()
    ref = g["%s/%s" % (tag, key)]
    assert y.shape == ref.shape
    assert float(np.abs(y - ref).max() / np.abs(ref).max()) < REL_TOL
    top2 = np.sort(ref, axis=-1)[..., -2:]
    clear = (top2[..., 1] - top2[..., 0]) > REL_TOL * np.abs(ref).max()
    assert (y.argmax(-1)[clear] == ref.argmax(-1)[clear]).all()


class _PieceCollate:
    """Stand-in for the reference's TokenCollateHF (which needs a HuggingFace tokenizer + LM): pieces are
    the text cut every 3 characters inside a word, 'embeddings' are a fixed random vector per piece."""

    def __init__(self, enc, ext):
        self._enc, self._ext, self._lang_id = enc, ext, 0

    def get_embeddings_size(self):
        return list(self._ext)

    def get_tokens(self, text):
        toks, ids, spa = [], [], []
        for word in text.split(" "):
            for i in range(0, len(word), 3):
                toks.append(word[i:i + 3])
                ids.append(hash(word[i:i + 3]) % 1000)
                spa.append(2 if i + 3 >= len(word) else 1)
        return toks, ids, spa

    def collate_fn(self, batch):
        from cube_amd import synth
        pieces, off, n_main, texts = [], [], [], []
        for ex in batch:
            seq = list(ex["prev"][0]) + list(ex["main"][0]) + list(ex["next"][0])
            pieces.append(seq)
            off.append(len(ex["prev"][0]))
            n_main.append(len(ex["main"][0]))
            texts.append(list(ex["main"][0]))
        X = synth.make_token_batch(self._enc, [len(p) for p in pieces], self._ext, seed=7, lang_ids=[self._lang_id] * len(batch))
        X["x_text"], X["y_offset"], X["y_len"] = texts, torch.tensor(off), torch.tensor(n_main)
        return X


def test_process_builds_the_document_the_labels_describe(dev):
    """Tokenizer.process: windows of 300 pieces with 100 of context (utils_tokenizer.py:58-112), arg-max
    labels of the main part of every window, decoded to sentences / tokens like tokenizer.py:258-304."""
    from oracle import tokenizer_oracle as to
    from cube_amd.networks.tokenizer import token_windows
    m, cfg, enc, sd = _model("full", dev)
    rng = np.random.default_rng(3)
    text = " ".join("".join(rng.choice(list("abcdefgh"), int(rng.integers(1, 9)))) for _ in range(400)) + "\n more  text"
    col = _PieceCollate(enc, CASES["full"]["ext"])
    doc = m.process(text, col, batch_size=2, lang_id=3)
    # the same labels through the oracle's decoder
    clean = " ".join(text.replace("\n", " ").split(" "))
    while "  " in clean:
        clean = clean.replace("  ", " ")
    ds = token_windows(clean, col.get_tokens)
    assert len(ds) >= 2 and all(len(e["main"][0]) <= 300 and len(e["prev"][0]) <= 100 for e in ds)
    toks, preds = [], []
    for s in range(0, len(ds), 2):
        b = col.collate_fn(ds[s:s + 2])
        y = m.forward(b).argmax(-1).cpu().numpy()
        for i in range(len(b["y_len"])):
            toks += b["x_text"][i]
            preds += y[i, int(b["y_offset"][i]):int(b["y_offset"][i]) + int(b["y_len"][i])].tolist()
    sents, mwes = to.decode_labels(toks, preds)
    assert [[w.word for w in s.words] for s in doc.sentences] == sents
    assert [[w.space_after.endswith(";compund") for w in s.words] for s in doc.sentences] == mwes
    assert all(s.lang_id == 3 for s in doc.sentences)
    assert "".join(w for s in sents for w in s) == clean.replace(" ", "")


def test_cube_api_raw_text_runs_tokenizer_then_parser(dev, tmp_path):
    """CubeObj with a caller-supplied sub-word collate: raw text -> Tokenizer.process -> lm_helper ->
    Parser.process (reference cube/api.py:117-125), == the stages by hand."""
    import yaml
    from cube_amd import synth
    from cube_amd.api import CubeObj
    from cube_amd.io_utils.config import ParserConfig
    from cube_amd.networks.lm import LMHelperDummy
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    tcase = cases.TOKENIZER_SMALL
    tcfg, enc, tsd = cases.tokenizer_model(tcase)
    pcfg = ParserConfig()
    for k, v in cases.TINY["dims"].items():
        setattr(pcfg, k, v)
    pcfg.lm_model = "dummy"
    psd = synth.make_parser_state_dict(pcfg, enc, [1], seed=5)
    base = str(tmp_path / "en")
    with open(base + "-trf-tokenizer.yaml", "w") as f:
        yaml.safe_dump({"language_codes": synth.EN_LANGUAGE_CODES[:3], "language_map": {"en": synth.EN_LANGUAGE_CODES[1]}}, f)
    for name, cfg, sd in (("tokenizer", tcfg, tsd), ("parser", pcfg, psd)):
        cfg.save("%s-trf-%s.config" % (base, name))
        enc.save("%s-trf-%s.encodings" % (base, name))
        torch.save({"state_dict": sd}, "%s-trf-%s.best" % (base, name))
    col = _PieceCollate(enc, tcase["ext"])
    cube = CubeObj(base, device=str(dev), lang="en", tokenizer_collate=col)
    text = "Abc defgh ij klmnopq rs tuv wxyz abcd efg"
    doc = cube(text)
    m, _, _, _ = _model("small", dev)
    ref = m.process(text, col, lang_id=1)
    LMHelperDummy().apply(ref)
    p = Parser(pcfg, enc, ext_word_emb=[1])
    p.load_state_dict(psd)
    p.to(dev)
    p.process(ref, MorphoCollate(enc), num_workers=0)
    assert len(doc.sentences) >= 1 and str(doc) == str(ref)
    assert "".join(w.word for s in doc.sentences for w in s.words) == text.replace(" ", "")
