"""External word-embedding providers that fill `word.emb` before the parser runs.

The real providers of the reference (fastText / XLM-R / Languasito, cube/networks/lm.py:34-211)
need model files that cannot be fetched offline and sit BEFORE the accelerated path (SURVEY.md
§8f rank 3); only the two model-free providers are mirrored here.
"""


class LMHelperDummy:
    """reference: cube/networks/lm.py:214-233 — one 1-dimensional layer, value 1.0."""

    def __init__(self, device="cpu", model=None):
        pass

    def get_embedding_size(self):
        return [1]

    def apply(self, document):
        for sent in document.sentences:
            for w in sent.words:
                w.emb = [[1.0]]

    def apply_raw(self, batch):
        return [[[1.0] for _ in sent] for sent in batch]


class LMHelperPrecomputed:
    """The caller has already set `word.emb` (list of K float vectors of the given sizes)."""

    def __init__(self, sizes):
        self._sizes = list(sizes)

    def get_embedding_size(self):
        return self._sizes

    def apply(self, document):
        for sent in document.sentences:
            for w in sent.words:
                if w.emb is None or len(w.emb) != len(self._sizes):
                    raise ValueError("word.emb must hold %d vectors before the parser runs" % len(self._sizes))
