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
        slen, wmax = hm.geometry(sents, 8)  # one C pass over the words, on the host thread pool (csrc/host_collate.c)
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


def _ids_of_words(sents, enc):
    """[n_words, 5] int32 (head, label id, upos id, xpos id, attrs id) read back from the words (the general
    way, for a parser object that does not report its result arrays)."""
    import numpy as np
    tabs = [{t: i for i, t in enumerate(lst)} for lst in (enc.labels, enc.upos_list, enc.xpos_list, enc.attrs_list)]
    rows = [(int(w.head), tabs[0][w.label], tabs[1][w.upos], tabs[2][w.xpos], tabs[3][w.attrs]) for s in sents for w in s.words]
    return np.asarray(rows, dtype=np.int32).reshape(-1, 5)


def _write_ids(sents, ids, enc):
    """the inverse: [n_words, 5] int32 -> head / label / upos / xpos / attrs of the words, in order."""
    from cube_amd import host
    hm = host.load()
    if hm is not None and isinstance(sents, list):
        hm.write_back_flat(sents, ids, (enc.labels, enc.upos_list, enc.xpos_list, enc.attrs_list))
        return
    k = 0
    for s in sents:
        for w in s.words:
            h, l, u, x, a = ids[k]
            w.head, w.label, w.upos, w.xpos, w.attrs = int(h), enc.labels[l], enc.upos_list[u], enc.xpos_list[x], enc.attrs_list[a]
            k += 1


def shard_result_ids(parser, doc, collate, lo, hi, batch_size=4, bucketed=False, max_tokens=8192):
    """Run the parser on sentences [lo, hi) of `doc` (results written into the words) and return them as one
    int32 [n_words_of_the_shard, 5] array (head, label id, upos id, xpos id, attrs id), words in document order.
    A cube_amd Parser hands over its result arrays batch by batch (`on_result`); any other object with the
    reference's `process` signature has them read back from the words."""
    import numpy as np
    sents = doc.sentences[lo:hi]
    wcount = np.fromiter((len(s.words) for s in sents), dtype=np.int64, count=len(sents))
    woff = np.concatenate([[0], np.cumsum(wcount)])
    ids = np.zeros((int(woff[-1]), 5), dtype=np.int32)
    if hi <= lo:
        return ids
    if not getattr(parser, "reports_results", False):
        parser.process(_Slice(doc, lo, hi), collate, batch_size=batch_size, num_workers=0)
        return _ids_of_words(sents, parser._encodings)

    def on_result(batch, where, res):
        # res int32 [5, B, L]; `where` = index of the batch's first sentence in the slice, or one index per sentence
        B, L = res.shape[1], res.shape[2]
        idx = np.arange(B) + int(where) if np.ndim(where) == 0 else np.asarray(where)
        n = wcount[idx]
        rows = np.repeat(np.arange(B), n)
        cols = np.arange(int(n.sum())) - np.repeat(np.cumsum(n) - n, n)
        ids[np.repeat(woff[idx], n) + cols] = res[:, rows, cols].T

    if bucketed:
        parser.process_bucketed(_Slice(doc, lo, hi), collate, max_tokens=max_tokens, on_result=on_result)
    else:
        parser.process(_Slice(doc, lo, hi), collate, batch_size=batch_size, num_workers=0, on_result=on_result)
    return ids


def gather_result_ids(ids, sizes, rank, world_size, device=None):
    """Gather every rank's [n_r, 5] int32 result array on rank 0 (the only traffic of the N>1 path,
    20 bytes per word).  `sizes` = words per rank, known to all ranks (every rank holds the whole document).
    One `dist.gather` of equal-size tensors padded to the largest shard — over RCCL when `device` is a GPU
    (the arrays then cross xGMI), over gloo on the CPU.  -> list of arrays on rank 0, None elsewhere."""
    import torch
    m = max(max(sizes), 1)
    t = torch.zeros((m, 5), dtype=torch.int32, device=device or "cpu")
    if len(ids):
        t[:len(ids)] = torch.from_numpy(ids).to(t.device)
    parts = [torch.empty_like(t) for _ in range(world_size)] if rank == 0 else None
    dist.gather(t, parts, dst=0)
    if rank != 0:
        return None
    return [p[:n].cpu().numpy() for p, n in zip(parts, sizes)]


def process_sharded(parser, doc, collate, batch_size=4, rank=None, world_size=None, gather=True,
                    lemmatizer=None, lemmatizer_collate=None, device=None, bucketed=False, max_tokens=8192):
    """Run `parser.process` on this rank's shard of `doc`; with gather=True rank 0 ends up with
    every word's head/label/upos/xpos/attrs filled (other ranks: their own shard only).  The results travel as
    int32 arrays (`gather_result_ids`), not as pickled strings.
    bucketed=True: the shard runs through `process_bucketed` (length-sorted batches of <= max_tokens, the
    throughput mode of SURVEY 8d config 3) instead of the reference's in-order batches.
    With a `lemmatizer`, the rank also lemmatises its shard (`process_bucketed`: one decode per distinct
    (word, UPOS, language), equal-length batches — a word's lemma does not depend on its batch, so the
    result does not depend on the sharding either); lemmas are strings and travel by `gather_object`."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    n = len(doc.sentences)
    bounds = [shard_bounds(n, r, world_size, batch_size) for r in range(world_size)]
    lo, hi = bounds[rank]
    ids = shard_result_ids(parser, doc, collate, lo, hi, batch_size, bucketed=bucketed, max_tokens=max_tokens)
    if hi > lo and lemmatizer is not None:
        lemmatizer.process_bucketed(_Slice(doc, lo, hi), lemmatizer_collate)
    if gather and world_size > 1:
        sizes = [sum(len(s.words) for s in doc.sentences[a:b]) for a, b in bounds]
        parts = gather_result_ids(ids, sizes, rank, world_size, device)
        if rank == 0:
            for r in range(1, world_size):
                a, b = bounds[r]
                if b > a:
                    _write_ids(doc.sentences[a:b], parts[r], parser._encodings)
        if lemmatizer is not None:
            mine = [[w.lemma for w in s.words] for s in doc.sentences[lo:hi]]
            got = [None] * world_size if rank == 0 else None
            dist.gather_object(mine, got, dst=0)
            if rank == 0:
                for r in range(1, world_size):
                    a, b = bounds[r]
                    for s, row in zip(doc.sentences[a:b], got[r]):
                        for w, lem in zip(s.words, row):
                            w.lemma = lem
    return doc
