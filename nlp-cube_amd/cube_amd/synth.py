"""Deterministic synthetic models and corpora.

No released checkpoint, treebank or embedding model can be fetched offline, so
every test, fixture and benchmark in this repo runs on a synthetic vocabulary,
a random-init `state_dict` with the reference's key names and shapes
(reference: cube/networks/parser.py:23-95, tagger.py:22-67, modules.py:424-448,
564-579) and synthetic CoNLL-U documents.  Everything is generated from
`numpy.random.default_rng(seed)` only, so the same objects can be re-created
bit-identically in the build container (where the reference is loaded with the
dict) and on the GPU box (where only this repo exists).
"""
import numpy as np
import torch

from cube_amd.io_utils.encodings import Encodings
from cube_amd.io_utils.objects import Document, Sentence, Word

# en flavours of the 3.0 'en' model (reference: scripts/train/2.5/language/en.yaml:7-22)
EN_LANGUAGE_CODES = ["en_esl", "en_pud", "en_gum", "en_ewt", "en_lines", "en_partut"]
EN_LANGUAGE_MAP = {"en": "en_ewt"}


def _symbols(n_chars):
    """n_chars-3 printable non-space symbols: lower-case letters first, then digits,
    punctuation, then letters from other scripts that have distinct upper/lower forms."""
    pool = [chr(c) for c in range(ord("a"), ord("z") + 1)] + [str(d) for d in range(10)]
    pool += list(".,;:!?-'\"()[]/&%$#@*+=<>_~^|\\{}`")
    pool += [chr(c) for c in range(0x3b1, 0x3ca) if chr(c).upper() != chr(c)]  # greek lower
    pool += [chr(c) for c in range(0x430, 0x450)]  # cyrillic lower
    out = []
    for s in pool:
        if s not in out and s != " ":
            out.append(s)
    if len(out) < n_chars - 3:
        raise ValueError("symbol pool too small")
    return out[: n_chars - 3]


def make_encodings(n_chars=100, n_words=20002, n_upos=19, n_xpos=52, n_attrs=240, n_labels=52, num_langs=6):
    """Synthetic vocabulary with the sizes of SURVEY.md §8d (all counts include <PAD>/<UNK>)."""
    chars = _symbols(n_chars)
    letters = [c for c in chars if c.upper() != c.lower()]
    rng = np.random.default_rng(20240)
    seen = set()
    words = []
    while len(words) < n_words - 2:  # lower-case strings, length 1+Poisson(4) clipped to [1,16]
        wl = int(np.clip(1 + rng.poisson(4), 1, 16))
        s = "".join(rng.choice(letters, wl))
        if s not in seen:
            seen.add(s)
            words.append(s)
    return Encodings.from_lists(chars, words,
                                ["U%d" % i for i in range(n_upos - 2)],
                                ["X%d" % i for i in range(n_xpos - 2)],
                                ["A%d=v" % i for i in range(n_attrs - 2)],
                                ["dep%d" % i for i in range(n_labels - 2)], num_langs)


# ------------------------------------------------------------------------------------------
# weights
# ------------------------------------------------------------------------------------------
def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class _Init:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)

    def normal(self, shape, std=1.0):
        return _t(self.rng.standard_normal(shape) * std)

    def uniform(self, shape, lo, hi):
        return _t(self.rng.uniform(lo, hi, shape))

    def xavier(self, shape, gain=1.0):
        # shape = [out, in] or [out, in, k]
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        return self.normal(shape, gain * np.sqrt(2.0 / (fan_in + fan_out)))

    def emb(self, rows, dim, padding_idx=None):
        w = self.normal((rows, dim))
        if padding_idx is not None:
            w[padding_idx] = 0
        return w

    def linear(self, sd, prefix, n_out, n_in, xavier=True):
        b = 1.0 / np.sqrt(n_in)
        sd[prefix + ".weight"] = self.xavier((n_out, n_in)) if xavier else self.uniform((n_out, n_in), -b, b)
        sd[prefix + ".bias"] = self.uniform((n_out,), -b, b)

    def gated_conv_stack(self, sd, prefix, n_layers, first_in, next_in, filters):
        cin = first_in
        for l in range(n_layers):
            p = "%s.%d" % (prefix, l)
            sd[p + ".0.conv.weight"] = self.xavier((filters, cin, 5), gain=5.0 / 3.0)
            b = 1.0 / np.sqrt(cin * 5)
            sd[p + ".0.conv.bias"] = self.uniform((filters,), -b, b)
            # BatchNorm1d in eval mode: non-trivial statistics so that folding is exercised
            sd[p + ".1.weight"] = self.uniform((filters,), 0.8, 1.2)
            sd[p + ".1.bias"] = self.normal((filters,), 0.1)
            sd[p + ".1.running_mean"] = self.normal((filters,), 0.1)
            sd[p + ".1.running_var"] = self.uniform((filters,), 0.5, 1.5)
            sd[p + ".1.num_batches_tracked"] = torch.tensor(100, dtype=torch.long)
            cin = next_in


def _common_front(ini, sd, config, enc, ext_word_emb, wordgram_chars):
    half_c = config.char_filter_size // 2
    half_w = config.cnn_filter // 2
    G = enc.num_langs
    # WordGram(num_chars, num_langs=G+1): tables num_chars+3 / G+2 / 4   (modules.py:430-432)
    sd["_word_net._lang_emb.weight"] = ini.emb(G + 2, config.lang_emb_size, 0)
    sd["_word_net._tok_emb.weight"] = ini.emb(wordgram_chars + 3, config.char_emb_size, 0)
    sd["_word_net._case_emb.weight"] = ini.emb(4, 32, 0)
    ini.gated_conv_stack(sd, "_word_net._convolutions_char", config.char_layers,
                         config.char_emb_size + 32 + config.lang_emb_size, half_c + config.lang_emb_size,
                         config.char_filter_size)
    ini.linear(sd, "_word_net._pre_out.linear_layer", half_c, half_c)
    sd["_zero_emb.weight"] = ini.emb(1, half_c)
    for i, e in enumerate(ext_word_emb):
        ini.linear(sd, "_ext_proj.%d._h1.linear_layer" % i, 500, e)
        ini.linear(sd, "_ext_proj.%d._h2.linear_layer" % i, config.external_proj_size, 500)
    sd["_word_emb.weight"] = ini.emb(len(enc.word2int), config.word_emb_size, 0)
    sd["_lang_emb.weight"] = ini.emb(G + 1, config.lang_emb_size, 0)
    first_in = half_c + config.lang_emb_size + config.word_emb_size + config.external_proj_size
    ini.gated_conv_stack(sd, "_convs", config.cnn_layers, first_in, half_w + config.lang_emb_size, config.cnn_filter)
    for name, n in (("_aupos", len(enc.upos2int)), ("_axpos", len(enc.xpos2int)), ("_aattrs", len(enc.attrs2int))):
        ini.linear(sd, name + ".linear_layer", n, half_c + config.lang_emb_size)
    return half_c, half_w, first_in


def make_parser_state_dict(config, enc, ext_word_emb, seed=1234):
    """Random `state_dict` for `Parser(config, enc, ext_word_emb=...)`; key names/shapes as
    SURVEY.md §8a-W.  The biaffine weights, zero-initialised by the reference
    (modules.py:541-542,555-556), are randomised so that arc/label parity is not vacuous."""
    if not isinstance(ext_word_emb, (list, tuple)):
        ext_word_emb = [ext_word_emb]
    ini = _Init(seed)
    sd = {}
    L = config.lang_emb_size
    half_c, half_w, first_in = _common_front(ini, sd, config, enc, ext_word_emb, len(enc.char2int) - 2)
    n_u, n_x, n_a, n_l = len(enc.upos2int), len(enc.xpos2int), len(enc.attrs2int), len(enc.label2int)
    ini.linear(sd, "_pre_morpho.linear_layer", half_w, half_w + L)
    ini.linear(sd, "_upos.linear_layer", n_u, half_w + L)
    ini.linear(sd, "_attrs.linear_layer", n_a, 64 + half_w + L)
    ini.linear(sd, "_xpos.linear_layer", n_x, 64 + half_w + L)
    sd["_upos_emb.weight"] = ini.emb(n_u, 64)
    H = config.rnn_size
    b = 1.0 / np.sqrt(H)
    for l in range(config.rnn_layers):
        nin = half_w + L + config.external_proj_size if l == 0 else 2 * H
        for suffix in ("", "_reverse"):
            sd["_rnn.weight_ih_l%d%s" % (l, suffix)] = ini.uniform((4 * H, nin), -b, b)
            sd["_rnn.weight_hh_l%d%s" % (l, suffix)] = ini.uniform((4 * H, H), -b, b)
            sd["_rnn.bias_ih_l%d%s" % (l, suffix)] = ini.uniform((4 * H,), -b, b)
            sd["_rnn.bias_hh_l%d%s" % (l, suffix)] = ini.uniform((4 * H,), -b, b)
    ini.linear(sd, "_pre_out.linear_layer", config.pre_parser_size, 2 * H + L)
    for name, hid, n_out in (("_head", config.head_size, 1), ("_label", config.label_size, n_l)):
        ini.linear(sd, name + ".W1", hid, config.pre_parser_size, xavier=False)
        ini.linear(sd, name + ".W2", hid, config.pre_parser_size, xavier=False)
    hs, ls = config.head_size + 1, config.label_size + 1
    sd["_head.scorer.W_bilin.weight"] = ini.normal((hs, hs, 1), 0.1)
    sd["_head.scorer.W_bilin.bias"] = ini.normal((1,), 0.1)
    sd["_label.scorer.W_bilin.weight"] = ini.normal((n_l, ls, ls), 0.05)
    sd["_label.scorer.W_bilin.bias"] = ini.normal((n_l,), 0.1)
    sd["_r_emb.weight"] = ini.emb(1, first_in)
    return sd


def make_tagger_state_dict(config, enc, ext_word_emb, seed=4321):
    """Random `state_dict` for the stand-alone `Tagger` (reference: tagger.py:22-67)."""
    if not isinstance(ext_word_emb, (list, tuple)):
        ext_word_emb = [ext_word_emb]
    ini = _Init(seed)
    sd = {}
    L = config.lang_emb_size
    half_c, half_w, first_in = _common_front(ini, sd, config, enc, ext_word_emb, len(enc.char2int))
    n_u, n_x, n_a = len(enc.upos2int), len(enc.xpos2int), len(enc.attrs2int)
    ini.linear(sd, "_upos.linear_layer", n_u, half_w + L)
    sd["_upos_emb.weight"] = ini.emb(n_u, 64)
    ini.linear(sd, "_xpos.linear_layer", n_x, 64 + L + half_w)
    ini.linear(sd, "_attrs.linear_layer", n_a, 64 + L + half_w)
    return sd


def make_lemmatizer_state_dict(config, enc, seed=2468, out_gain=10.0, rec_gain=3.0, eol_gain=3.0):
    """Random `state_dict` for `Lemmatizer(config, enc)` (reference: lemmatizer.py:24-69,
    modules.py:133-143).  A default-initialised random decoder sits in a fixed point and prints
    one character until the 20 x length failsafe (lemmatizer.py:141); the gains give the decoder
    LSTM livelier dynamics (rec_gain), logits with clear margins (out_gain) and an end-of-lemma
    logit that wins every few steps (eol_gain), so that a batch mixes words that stop after a few
    characters with words that run into the failsafe."""
    ini = _Init(seed)
    sd = {}
    G, C = enc.num_langs, len(enc.char2int)
    F_, half = 512, 256  # NUM_FILTERS is a constant of the reference (lemmatizer.py:26)
    le, ue, ce, D = config.lang_emb_size, config.upos_emb_size, config.char_emb_size, config.decoder_size
    sd["_lang_emb.weight"] = ini.emb(G + 1, le, 0)
    sd["_upos_emb.weight"] = ini.emb(len(enc.upos2int), ue, 0)
    sd["_char_emb.weight"] = ini.emb(C + 2, ce, 0)
    sd["_case_emb.weight"] = ini.emb(4, 16, 0)
    ini.gated_conv_stack(sd, "_convolutions_char", 5, ce + le + ue + 16, half + le, F_)
    b = 1.0 / np.sqrt(D)
    frame = half + ce + le + ue + 16
    for l in range(config.decoder_layers):
        nin = frame if l == 0 else D
        sd["_decoder.weight_ih_l%d" % l] = ini.uniform((4 * D, nin), -b, b) * rec_gain
        sd["_decoder.weight_hh_l%d" % l] = ini.uniform((4 * D, D), -b, b) * rec_gain
        sd["_decoder.bias_ih_l%d" % l] = ini.uniform((4 * D,), -b, b)
        sd["_decoder.bias_hh_l%d" % l] = ini.uniform((4 * D,), -b, b)
    att_in = (half + le + ue + 16) // 2 * 2 + D
    sd["_attention.attn.conv.weight"] = ini.xavier((config.att_proj_size, att_in, 5), gain=5.0 / 3.0)
    ba = 1.0 / np.sqrt(att_in * 5)
    sd["_attention.attn.conv.bias"] = ini.uniform((config.att_proj_size,), -ba, ba)
    sd["_attention.v"] = ini.uniform((config.att_proj_size,), 0.0, 1.0)
    ini.linear(sd, "_output_char.linear_layer", C + 2, D)
    ini.linear(sd, "_output_case.linear_layer", 4, D)
    sd["_output_char.linear_layer.weight"] *= out_gain
    sd["_output_char.linear_layer.weight"][C] *= eol_gain
    # the stop index C+1 has no entry in `encodings.characters`: a trained model never emits it, and
    # the reference would raise IndexError if it did (lemmatizer.py:217)
    sd["_output_char.linear_layer.bias"][C + 1] -= 10.0
    sd["_start_frame.weight"] = ini.emb(1, frame)
    return sd


def make_tokenizer_state_dict(config, enc, ext_word_emb, seed=1357):
    """Random `state_dict` for `Tokenizer(config, enc, ext_word_emb=...)` (reference: tokenizer.py:26-59;
    its WordGram keeps the class defaults: 3 layers, lang_emb 32, modules.py:425-426)."""
    if not isinstance(ext_word_emb, (list, tuple)):
        ext_word_emb = [ext_word_emb]
    ini = _Init(seed)
    sd = {}
    G, C = enc.num_langs, len(enc.char2int)
    half, le, P = config.cnn_filter // 2, config.lang_emb_size, config.external_proj_size
    ini.gated_conv_stack(sd, "_convs", config.cnn_layers, P + le + 256 + 16, half + le, config.cnn_filter)
    sd["_wg._lang_emb.weight"] = ini.emb(G + 1, 32, 0)
    sd["_wg._tok_emb.weight"] = ini.emb(C - 2 + 3, 256, 0)
    sd["_wg._case_emb.weight"] = ini.emb(4, 32, 0)
    ini.gated_conv_stack(sd, "_wg._convolutions_char", 3, 256 + 32 + 32, 256 + 32, 512)
    ini.linear(sd, "_wg._pre_out.linear_layer", 256, 256)
    sd["_lang_emb.weight"] = ini.emb(G + 1, le, 0)
    sd["_spa_emb.weight"] = ini.emb(3, 16, 0)
    ini.linear(sd, "_output.linear_layer", 5, half + le)
    for i, e in enumerate(ext_word_emb):
        ini.linear(sd, "_ext_proj.%d._h1.linear_layer" % i, 500, e)
        ini.linear(sd, "_ext_proj.%d._h2.linear_layer" % i, P, 500)
    return sd


def make_token_batch(enc, sent_pieces, ext_word_emb, seed=0, lang_ids=None):
    """A batch in the format `TokenCollateHF.collate_fn` hands to `Tokenizer.forward`
    (utils_tokenizer.py:394-576), built from synthetic sub-word pieces instead of a HuggingFace
    tokenizer + LM (not available offline): `sent_pieces` = pieces per example; x_input = K tensors
    [B, T, E_k] ~ N(0,1) with zero rows at the pads (the LM output at pad positions is replaced by
    zeros here; the reference feeds whatever the LM produced for its PAD id)."""
    rng = np.random.default_rng(seed)
    chars = [c for c, i in enc.char2int.items() if i >= 3]
    B, T = len(sent_pieces), int(max(sent_pieces))
    words, langs = [], []
    x_lang = np.zeros(B, dtype=np.int64)
    for b, n in enumerate(sent_pieces):
        lang = int(rng.integers(enc.num_langs)) if lang_ids is None else int(lang_ids[b])
        x_lang[b] = lang + 1
        for _ in range(int(n)):
            wl = int(np.clip(1 + rng.poisson(3), 1, 12))
            w = "".join(rng.choice(chars, wl))
            if rng.random() < 0.2:
                w = w[0].upper() + w[1:]
            words.append(w)
            langs.append(lang)   # the reference passes the raw language id here, not id + 1 (:427)
    N, W = len(words), max(len(w) for w in words)
    x_char = np.zeros((N, W), dtype=np.int64)
    x_case = np.zeros((N, W), dtype=np.int64)
    x_mask = np.zeros((N, W), dtype=np.float64)
    for i, w in enumerate(words):
        for p, ch in enumerate(w):
            x_mask[i, p] = 1
            x_case[i, p] = 1 if ch.lower() == ch.upper() else (2 if ch.lower() != ch else 3)
            x_char[i, p] = enc.char2int.get(ch.lower(), enc.char2int["<UNK>"])
    spa = np.zeros((B, T), dtype=np.int64)
    x_input = []
    valid = np.arange(T)[None, :] < np.asarray(sent_pieces)[:, None]
    spa[valid] = rng.integers(1, 3, size=int(valid.sum()))
    for e in ext_word_emb:
        x = rng.standard_normal((B, T, e)).astype(np.float32)
        x[~valid] = 0
        x_input.append(torch.from_numpy(x))
    text, pos = [], 0
    for n in sent_pieces:
        text.append(words[pos:pos + int(n)])
        pos += int(n)
    return {"x_input": x_input, "x_input_spa": torch.from_numpy(spa), "x_word_char": torch.from_numpy(x_char),
            "x_word_case": torch.from_numpy(x_case), "x_word_masks": torch.from_numpy(x_mask),
            "x_word_len": torch.tensor([len(w) for w in words], dtype=torch.long),
            "x_word_lang": torch.tensor(langs, dtype=torch.long), "x_text": text, "x_lang": torch.from_numpy(x_lang),
            "y_offset": torch.zeros(B, dtype=torch.long), "y_len": torch.tensor([int(n) for n in sent_pieces]),
            "x_sent_len": torch.tensor([int(n) for n in sent_pieces], dtype=torch.long)}


# ------------------------------------------------------------------------------------------
# corpora
# ------------------------------------------------------------------------------------------
def make_document(enc, sent_lens, seed=0, ext_word_emb=(300,), lang_ids=None, unk_rate=0.3,
                  max_word_len=16, emb_dtype=np.float32):
    """Synthetic tokenised document: word length 1+Poisson(4) clipped to
    [1,max_word_len]; characters uniform over the known symbols; 15 % capitalised, 10 %
    symbol-only, `unk_rate` of the words out of vocabulary (the rest are vocabulary
    entries, whose spelling 'w<i>' then also fixes their characters); `word.emb` = K
    vectors ~ N(0,1) in float32.  `lang_ids` = one id per sentence (default: uniform)."""
    rng = np.random.default_rng(seed)
    chars = [c for c, i in enc.char2int.items() if i >= 3]
    letters = [c for c in chars if c.upper() != c.lower()]
    symbols = [c for c in chars if c.upper() == c.lower()]
    vocab = [w for w, i in enc.word2int.items() if i >= 2]
    doc = Document()
    for si, n in enumerate(sent_lens):
        rows = []
        for wi in range(int(n)):
            u = rng.random()
            if u < unk_rate or not vocab:
                wl = int(np.clip(1 + rng.poisson(4), 1, max_word_len))
                kind = rng.random()
                if kind < 0.10:
                    s = "".join(rng.choice(symbols, wl))
                else:
                    s = "".join(rng.choice(letters, wl))
                    if kind < 0.25:
                        s = s[0].upper() + s[1:]
                    elif kind < 0.28:
                        s = s.upper()
            else:
                s = vocab[int(rng.integers(len(vocab)))]
                if rng.random() < 0.15:
                    s = s[0].upper() + s[1:]
            w = Word(wi + 1, s, "_", "_", "_", "_", 0, "_", "_", "_")
            w.emb = [rng.standard_normal(e).astype(emb_dtype) for e in ext_word_emb]
            rows.append(w)
        lang = int(rng.integers(enc.num_langs)) if lang_ids is None else int(lang_ids[si])
        doc.sentences.append(Sentence(sequence=rows, lang_id=lang))
    return doc


def zipf_words(doc, pool_size=20000, seed=5, one_language=True):
    """Re-draw every word of `doc` from a pool of its own first `pool_size` distinct words with Zipf frequencies
    (p_k ~ 1/k), as text has them: most tokens of a batch then repeat, which the all-new words of `make_document` never
    do.  The embeddings stay per token.  In place; -> doc."""
    pool = list(dict.fromkeys(w.word for s in doc.sentences for w in s.words))[:pool_size]
    p = 1.0 / np.arange(1, len(pool) + 1)
    rng = np.random.default_rng(seed)
    draw = rng.choice(len(pool), size=sum(len(s.words) for s in doc.sentences), p=p / p.sum())
    i = 0
    for s in doc.sentences:
        if one_language:
            s.lang_id = doc.sentences[0].lang_id
        for w in s.words:
            w.word = pool[draw[i]]
            i += 1
    return doc
