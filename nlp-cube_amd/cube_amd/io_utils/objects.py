"""CoNLL-U data model handed to / returned by the parser path.

Host-side mirror of the reference's document objects (reference:
cube/io_utils/objects.py:7-177).  The parser path only needs a small part of
it: `Sentence.words`, `Sentence.lang_id`, `Word.word`, `Word.emb` as inputs and
`Word.head/label/upos/xpos/attrs` as outputs, plus the CoNLL-U text rendering
used by the parity tests (`str(doc)` must be byte-identical to the reference's).
"""
import gzip


def _open_text(path, mode="r"):
    # reference: cube/io_utils/misc.py:4-10 (gz-aware open)
    if path.endswith(".gz"):
        return gzip.open(path, mode + "t", encoding="utf-8")
    return open(path, mode, encoding="utf-8")


def _index_or_range(value):
    """-> (parsed, is_compound).  '3' -> (3, False); '3-4' -> ('3-4', True).
    reference: cube/io_utils/objects.py:147-151"""
    try:
        return int(value), False
    except ValueError:
        return value, True


class Word:
    """One CoNLL-U row (reference: cube/io_utils/objects.py:154-177)."""

    __slots__ = ("index", "is_compound_entry", "word", "lemma", "upos", "xpos", "attrs", "head",
                 "label", "deps", "space_after", "parent", "emb", "token")

    def __init__(self, index, word, lemma, upos, xpos, attrs, head, label, deps, space_after, token=None):
        self.index, self.is_compound_entry = _index_or_range(index)
        self.word = word
        self.lemma = lemma
        self.upos = upos
        self.xpos = xpos
        self.attrs = attrs
        self.head, _ = _index_or_range(head)
        self.label = label
        self.deps = deps
        self.space_after = space_after
        self.parent = token
        self.token = token
        self.emb = None

    def __repr__(self):
        cols = (str(self.index), self.word, self.lemma, self.upos, self.xpos, self.attrs,
                str(self.head), self.label, self.deps, self.space_after)
        return "\t".join(c if isinstance(c, str) else c.encode("utf-8") for c in cols)


class Token:
    """Surface token = one or more words (reference: cube/io_utils/objects.py:121-144)."""

    def __init__(self, index=0, text="", words=None, space_after=True):
        self.index = index
        self.text = text
        self.words = [] if words is None else words
        self.other = None
        self.space_after = space_after

    def __repr__(self):
        rows = [str(w) for w in self.words]
        if len(self.words) > 1:
            misc = "_" if self.space_after else "SpaceAfter=No"
            rows.insert(0, "\t".join([str(self.index), self.text] + ["_"] * 7 + [misc]))
        return "\n".join(rows)


class Sentence:
    """reference: cube/io_utils/objects.py:71-118.  Rows whose index contains '.'
    (empty nodes) are dropped; a compound row 'a-b' opens a token that swallows
    the next b-a+1 words."""

    def __init__(self, sequence=None, lang_id=0, text=None):
        self.doc = None
        self.tokens = []
        self.words = []
        self.lang_id = lang_id
        pending = 0
        tok = None
        for w in (sequence or []):
            if "." in str(w.index):
                continue
            if pending == 0:
                tok = Token(index=w.index, text=w.word,
                            space_after=("spaceafter=no" not in w.space_after.lower()), words=[])
                self.tokens.append(tok)
                if w.is_compound_entry:
                    lo, hi = w.index.split("-")
                    pending = int(hi) - int(lo) + 1
                else:
                    pending = 1
            if not w.is_compound_entry:
                pending -= 1
                w.token = tok
                tok.words.append(w)
                self.words.append(w)
        self.text = self._detokenize() if text is None else text

    def _detokenize(self):
        return "".join(t.text + (" " if t.space_after else "") for t in self.tokens)

    def __repr__(self):
        return "\n".join(str(t) for t in self.tokens)


class Document:
    """reference: cube/io_utils/objects.py:7-68."""

    def __init__(self, filename=None, lang_id=0):
        self.sentences = []
        if filename is not None:
            self.load(filename, lang_id)

    def load(self, filename, lang_id=0):
        rows = []
        inside = False
        with _open_text(filename, "r") as f:
            for line in f.readlines():
                line = line.replace("\n", "").replace("\r", "")
                if line == "":
                    inside = False
                    if rows:
                        self.sentences.append(Sentence(sequence=rows, lang_id=lang_id))
                    rows = []
                elif inside or not line.startswith("#"):
                    p = line.split("\t")
                    rows.append(Word(p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8], p[9]))
                    inside = True

    def text(self):
        return self.__repr__()

    def __repr__(self):
        return "\n\n".join(str(s) for s in self.sentences)
