"""Seeded model / document / MST cases shared by the golden generator (tests/golden/
make_golden.py, build container only) and by the parity tests (which re-create the same
inputs from the same seeds and compare with the stored reference outputs)."""
import numpy as np

from cube_amd import synth
from cube_amd.io_utils.config import ParserConfig, TaggerConfig, LemmatizerConfig, TokenizerConfig

# tiny dims: every stage exercised, weights small, reference runs in milliseconds
TINY = dict(name="tiny", seed=11, ext=(12, 8), dims=dict(char_emb_size=16, char_filter_size=32, char_layers=3,
            word_emb_size=24, lang_emb_size=8, cnn_filter=32, cnn_layers=4, aux_softmax_location=3,
            pre_parser_size=40, head_size=12, label_size=20, external_proj_size=3, rnn_size=16, rnn_layers=2),
            vocab=dict(n_chars=60, n_words=300, n_upos=9, n_xpos=14, n_attrs=20, n_labels=11, num_langs=3))
# full dims = default ParserConfig, SURVEY.md §8d vocabulary
FULL = dict(name="full", seed=1234, ext=(300,), dims={}, vocab={})
# SURVEY §8d secondary input: 13 LM layers of 768 values per word ("transformer-like", lm.py:141-208)
FULL_K13 = dict(name="full_k13", seed=1234, ext=(768,) * 13, dims={}, vocab={})
TINY_TAGGER = dict(name="tiny_tagger", seed=12, ext=(12,), dims=dict(char_emb_size=16, char_filter_size=32,
                   char_layers=2, word_emb_size=24, lang_emb_size=8, cnn_filter=32, cnn_layers=3,
                   external_proj_size=5), vocab=TINY["vocab"])
FULL_TAGGER = dict(name="full_tagger", seed=4321, ext=(300,), dims={}, vocab={})

_DOCS = {
    # key: (sentence lengths, seed)
    "g1": ([5, 1, 9, 3, 12, 2, 7, 7, 4, 1, 6, 10, 3, 8, 2, 5], 101),
    "equal": ([6] * 5, 102),
    "ragged": ([9, 2, 14, 1, 5, 7], 103),
    "single": ([1], 104),
    "full_8x32": ([32] * 8, 105),
    "full_2x128": ([128] * 2, 106),
    "g6": (None, 107),  # 32 sentences, lengths 1..24
}


def _cfg(cls, dims):
    c = cls()
    for k, v in dims.items():
        setattr(c, k, v)
    return c


def parser_model(case):
    cfg = _cfg(ParserConfig, case["dims"])
    enc = synth.make_encodings(**case["vocab"])
    sd = synth.make_parser_state_dict(cfg, enc, list(case["ext"]), seed=case["seed"])
    return cfg, enc, sd


def tagger_model(case):
    cfg = _cfg(TaggerConfig, case["dims"])
    enc = synth.make_encodings(**case["vocab"])
    sd = synth.make_tagger_state_dict(cfg, enc, list(case["ext"]), seed=case["seed"])
    return cfg, enc, sd


LEMMA = dict(name="lemma", seed=2468, ext=(), dims={}, vocab={})
# smaller embeddings / decoder (NUM_FILTERS = 512 is fixed by the reference, lemmatizer.py:26)
LEMMA_SMALL = dict(name="lemma_small", seed=97, ext=(), dims=dict(decoder_size=48, att_proj_size=20, upos_emb_size=12,
                   lang_emb_size=8, char_emb_size=20), vocab=TINY["vocab"])


def lemmatizer_model(case):
    cfg = _cfg(LemmatizerConfig, case["dims"])
    enc = synth.make_encodings(**case["vocab"])
    sd = synth.make_lemmatizer_state_dict(cfg, enc, seed=case["seed"])
    return cfg, enc, sd


TOKENIZER = dict(name="tokenizer", seed=1357, ext=(48, 48, 48), dims={}, vocab={})
TOKENIZER_SMALL = dict(name="tokenizer_small", seed=1358, ext=(12,), dims=dict(cnn_filter=64, lang_emb_size=8, cnn_layers=3,
                       external_proj_size=3), vocab=TINY["vocab"])
_TOKEN_BATCHES = {"ragged": ([9, 2, 14, 1, 5, 7], 201), "equal": ([11] * 5, 202), "one": ([1], 203),
                  "long": ([300, 120, 257], 204)}


def tokenizer_model(case):
    cfg = _cfg(TokenizerConfig, case["dims"])
    enc = synth.make_encodings(**case["vocab"])
    sd = synth.make_tokenizer_state_dict(cfg, enc, list(case["ext"]), seed=case["seed"])
    return cfg, enc, sd


def token_batch(case, enc, key):
    pieces, seed = _TOKEN_BATCHES[key]
    return synth.make_token_batch(enc, pieces, case["ext"], seed=seed)


def lemma_document(enc, key, max_word_len=16):
    """document for the lemmatizer: words as `document`, plus a UPOS per word (what the parser
    would have written; some '_' = unknown tag) and no external embeddings."""
    lens, seed = _DOCS[key]
    if lens is None:
        lens = list(np.random.default_rng(seed).integers(1, 25, size=32))
    doc = synth.make_document(enc, lens, seed=seed, ext_word_emb=(), max_word_len=max_word_len)
    rng = np.random.default_rng(seed + 5000)
    tags = [t for t in enc.upos_list if t not in ("<PAD>",)] + ["_"]
    for s in doc.sentences:
        for w in s.words:
            w.upos = tags[int(rng.integers(len(tags)))]
    return doc


def document(case, enc, key):
    lens, seed = _DOCS[key]
    if lens is None:
        lens = list(np.random.default_rng(seed).integers(1, 25, size=32))
    return synth.make_document(enc, lens, seed=seed, ext_word_emb=case["ext"])


# ------------------------------------------------------------------------------------------
# MST cases (G4): arc probability tensors [B, L, L+1] + lens
# ------------------------------------------------------------------------------------------
def _softmax(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return (e / e.sum(axis=-1, keepdims=True)).astype(np.float32)


def mst_cases():
    out = {}
    for L in (1, 2, 3, 5, 8, 16, 32, 64, 128):
        for sigma in (0.5, 3.0, 8.0):
            rng = np.random.default_rng(1000 + L * 10 + int(sigma))
            B = 24 if L <= 32 else (8 if L == 64 else 4)
            out["rand_L%d_s%g" % (L, sigma)] = (_softmax(rng.standard_normal((B, L, L + 1)) * sigma),
                                                 np.full(B, L, dtype=np.int64))
    # ragged lens with padded columns carrying probability mass (softmax over all Lmax+1 columns)
    rng = np.random.default_rng(7)
    L = 20
    lens = rng.integers(1, L + 1, size=32).astype(np.int64)
    lens[0] = L
    out["ragged_L20"] = (_softmax(rng.standard_normal((32, L, L + 1)) * 3), lens)
    rng = np.random.default_rng(8)
    L = 48
    lens = rng.integers(1, L + 1, size=12).astype(np.int64)
    lens[0] = L
    out["ragged_L48"] = (_softmax(rng.standard_normal((12, L, L + 1)) * 2), lens)
    # uniform rows: every argmax is a tie (first index wins)
    for L in (4, 9):
        out["uniform_L%d" % L] = (np.full((3, L, L + 1), 1.0 / (L + 1), dtype=np.float32), np.full(3, L, np.int64))
    # forced cycles: 2-cycles and a 3-cycle with weak root scores
    L = 6
    p = np.full((4, L, L + 1), 0.01, dtype=np.float32)
    p[0, 0, 2] = p[0, 1, 1] = 0.9   # words 1<->2
    p[0, 2, 4] = p[0, 3, 3] = 0.8   # words 3<->4
    p[0, 4, 6] = p[0, 5, 5] = 0.7   # words 5<->6
    p[1, 0, 2] = p[1, 1, 3] = p[1, 2, 1] = 0.9  # 1->2->3->1
    p[1, 3, 0] = 0.5
    p[2, :, 0] = 0.6                # everybody wants the root (multi-root repair)
    p[3, 0, 2] = p[3, 1, 1] = 0.95  # cycle + nobody likes the root (zero-root repair)
    p[3, :, 0] = 0.001
    out["cycles_L6"] = (p, np.full(4, L, np.int64))
    # nested contractions: chain of mutual preferences
    L = 12
    rng = np.random.default_rng(9)
    p = _softmax(rng.standard_normal((6, L, L + 1)))
    for b in range(6):
        for i in range(0, L - 1, 2):
            p[b, i, i + 2] += 2.0 + b
            p[b, i + 1, i + 1] += 2.0 + b
    p /= p.sum(-1, keepdims=True)
    out["nested_L12"] = (p.astype(np.float32), np.full(6, L, np.int64))
    # exact zeros inside rows (tree_probs.all() == False paths) but still decodable
    L = 5
    p = _softmax(np.random.default_rng(10).standard_normal((5, L, L + 1)) * 2)
    p[0, :, 0] = 0.0
    p[0, 2, 0] = 0.3
    p[1, 1, :] = 0.0
    p[1, 1, 3] = 1.0
    p[2, 3, 1:] = 0.0
    p[3, 0, 1:] = 0.0
    p[4, :, 2] = 0.0
    out["zeros_L5"] = (p.astype(np.float32), np.full(5, L, np.int64))
    # no word may attach to the root with non-zero probability: the reference asserts
    # (utils.py:556-562) -> expected "no tree" failure
    p = _softmax(np.random.default_rng(12).standard_normal((1, L, L + 1)) * 2)
    p[0, :, 0] = 0.0
    out["notree_L5"] = (p.astype(np.float32), np.full(1, L, np.int64))
    return out


# ---- external-embedding stage: stand-ins for the tokenizer / transformer objects that LMHelperHF holds
# (lm.py:122-135; the real ones need model files that cannot be fetched).  Deterministic from seeds, so that the
# reference's own `_compute_we` / `apply` (tests/golden/make_golden.py g12) and this repo's see the same hidden states.
LM_LAYERS, LM_WIDTH, LM_PIECES = 13, 768, 1000   # lm.py:193,206-207 hard-code 13 x 768


class FakeSplitter:
    """`splitter(word)['input_ids']` = [START] + pieces + [END] (lm.py:155): 0..5 pieces per word from a CRC of the word."""

    def __call__(self, word):
        import zlib
        h = zlib.crc32(word.encode("utf-8"))
        n = 0 if word.startswith("\u200b") else 1 + (h % 5 if h % 7 else 0)
        return {"input_ids": [0] + [3 + (h >> (3 * j)) % (LM_PIECES - 3) for j in range(n)] + [2]}


class FakeEncoder:
    """`model(x, return_dict=True)['hidden_states']`: K tensors [B, len, E], value = table_k[id] + 0.1 * pos_k[position in
    the window it was called with] -- like the real model, a window's output depends on the window alone (lm.py:177-181)."""

    def __init__(self, seed=7, device="cpu"):
        import torch
        g = torch.Generator().manual_seed(seed)
        self.tok = [torch.randn((LM_PIECES, LM_WIDTH), generator=g) for _ in range(LM_LAYERS)]
        self.pos = [torch.randn((512, LM_WIDTH), generator=g) for _ in range(LM_LAYERS)]
        self.device = device

    def eval(self):
        return self

    def to(self, device):
        self.tok = [t.to(device) for t in self.tok]
        self.pos = [t.to(device) for t in self.pos]
        self.device = device
        return self

    def __call__(self, x, return_dict=True):
        hs = tuple(t[x] + 0.1 * p[:x.shape[1]].unsqueeze(0) for t, p in zip(self.tok, self.pos))
        return {"hidden_states": hs}


def lm_document(enc):
    """three short sentences, one of a single word, one word without pieces, and a sentence of 220 words whose ~580 pieces
    cross the 512-piece window of lm.py:171-181"""
    doc = synth.make_document(enc, [5, 1, 12, 220], seed=4242, ext_word_emb=(4,))
    doc.sentences[2].words[3].word = "\u200bnothing"
    for s in doc.sentences:
        for w in s.words:
            w.emb = None
    return doc
