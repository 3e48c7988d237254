"""Sentence sharding across the GPUs of one node.

Sentences are independent, so the N>1 path is: one process per GPU, weights replicated, rank r
runs `Parser.process` on its contiguous block of sentences, no collective on the data path.
Results depend on which sentences share a batch (the reference pads to the batch maximum and its
convolutions / LSTM see the pad positions, parser.py:201), so shard boundaries are multiples of
`batch_size`: every rank then forms exactly the batches the single-process reference would.
The optional result gather moves five small integer/string columns per word to rank 0.
"""
import torch.distributed as dist


def shard_bounds(n_sentences, rank, world_size, batch_size):
    """[lo, hi) of rank's sentences: whole batches, contiguous, in order, balanced to one batch."""
    n_batches = (n_sentences + batch_size - 1) // batch_size
    per, extra = divmod(n_batches, world_size)
    first = rank * per + min(rank, extra)
    count = per + (1 if rank < extra else 0)
    return min(first * batch_size, n_sentences), min((first + count) * batch_size, n_sentences)


def bucketed_batches(sent_lens, max_tokens=8192, max_sentences=1024):
    """Length-bucketed batches: sentence indices sorted by length (stable), cut greedily so that
    sentences x longest length <= max_tokens.  -> list of index arrays (each sorted by length)."""
    import numpy as np
    lens = np.asarray(sent_lens, dtype=np.int64)
    order = np.argsort(lens, kind="stable")
    out, start = [], 0
    while start < len(order):
        end = start + 1
        while end < len(order) and end - start < max_sentences and (end - start + 1) * int(lens[order[end]]) <= max_tokens:
            end += 1
        out.append(order[start:end])
        start = end
    return out


def merged_reference_batches(keys, batch_size, cap):
    """The reference's in-order batches (`batch_size` consecutive items each), merged into device
    batches WITHOUT changing any result.  What an item's result depends on in its reference batch is
    only the padding geometry of that batch (`keys[i]` = the geometry of reference batch i: longest
    sentence and longest word for the parser / tagger, longest word for the lemmatizer): items never
    interact otherwise.  Reference batches with equal geometry are therefore concatenated, up to `cap`
    items per device batch (always whole reference batches), and every item sees exactly the padding
    it would have seen alone with its own batch.
    keys: one hashable per reference batch; cap: items per device batch, an int or a function of the
    key -> list of lists of reference-batch indices (buckets in order of first appearance)."""
    buckets = {}
    for i, k in enumerate(keys):
        buckets.setdefault(k, []).append(i)
    out = []
    for k, idx in buckets.items():
        per = max(1, (cap(k) if callable(cap) else cap) // max(batch_size, 1))
        for s in range(0, len(idx), per):
            out.append(idx[s:s + per])
    return out


def sentence_batch_keys(sents, starts, batch_size, cap):
    """Padding geometry (longest sentence, longest word) of each reference batch of sentences, for
    `merged_reference_batches`.  The longest word costs a pass over every word, so it is only looked
    up for batches that can merge at all (`cap(key) >= 2 * batch_size`, which the longest sentence
    decides); the others get a key of their own."""
    from cube_amd import host
    hm = host.load()
    if hm is not None and isinstance(sents, list) and len(starts) > 1 and \
            all(b - a == batch_size for a, b in zip(starts[:-1], starts[1:])) and starts[0] == 0:
        import numpy as np
        slen, wmax = hm.geometry(sents)  # one C pass over the words (csrc/host_collate.c)
        idx = np.asarray(starts, dtype=np.int64)
        lmax = np.maximum.reduceat(slen, idx)
        wm = np.maximum.reduceat(wmax, idx)
        return [(int(l), int(w)) if cap((int(l), 0)) >= 2 * batch_size else (int(l), -1 - a)
                for l, w, a in zip(lmax, wm, starts)]
    keys = []
    for a in starts:
        group = sents[a:a + batch_size]
        lmax = max(len(s.words) for s in group)
        if cap((lmax, 0)) < 2 * batch_size:
            keys.append((lmax, -1 - a))
        else:
            keys.append((lmax, max((len(w.word) for s in group for w in s.words), default=0)))
    return keys


class _Slice:
    """A view of a Document holding only sentences[lo:hi] (same Sentence objects)."""

    def __init__(self, doc, lo, hi):
        self.sentences = doc.sentences[lo:hi]


def process_sharded(parser, doc, collate, batch_size=4, rank=None, world_size=None, gather=True,
                    lemmatizer=None, lemmatizer_collate=None):
    """Run `parser.process` on this rank's shard of `doc`; with gather=True rank 0 ends up with
    every word's head/label/upos/xpos/attrs filled (other ranks: their own shard only).
    With a `lemmatizer`, the rank also lemmatises its shard (`process_bucketed`: one decode per distinct
    (word, UPOS, language), equal-length batches — a word's lemma does not depend on its batch, so the
    result does not depend on the sharding either) and the lemma column travels with the others."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    lo, hi = shard_bounds(len(doc.sentences), rank, world_size, batch_size)
    if hi > lo:
        parser.process(_Slice(doc, lo, hi), collate, batch_size=batch_size, num_workers=0)
        if lemmatizer is not None:
            lemmatizer.process_bucketed(_Slice(doc, lo, hi), lemmatizer_collate)
    if gather and world_size > 1:
        mine = [[(int(w.head), w.label, w.upos, w.xpos, w.attrs, w.lemma) for w in s.words] for s in doc.sentences[lo:hi]]
        parts = [None] * world_size if rank == 0 else None
        dist.gather_object((lo, hi, mine), parts, dst=0)
        if rank == 0:
            for plo, phi, rows in parts:
                for s, srow in zip(doc.sentences[plo:phi], rows):
                    for w, (h, lab, u, x, a, lem) in zip(s.words, srow):
                        w.head, w.label, w.upos, w.xpos, w.attrs, w.lemma = h, lab, u, x, a, lem
    return doc
