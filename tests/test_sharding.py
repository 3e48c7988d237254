"""N>1 path on CPU: world_size-2 gloo run of the sentence sharding + result gather, with a
stand-in parser (the HIP parser needs a GPU; the sharding logic does not)."""
import os
import sys

import torch.distributed as dist
import torch.multiprocessing as mp

from cube_amd import synth
from cube_amd.sharding import shard_bounds, process_sharded


def test_shard_bounds_cover_every_sentence_once_in_whole_batches():
    for n in (0, 1, 7, 32, 33, 100, 1001):
        for world in (1, 2, 3, 8):
            for bs in (1, 4, 32):
                seen = []
                for r in range(world):
                    lo, hi = shard_bounds(n, r, world, bs)
                    assert lo % bs == 0 or lo == n
                    seen += list(range(lo, hi))
                assert seen == list(range(n)), (n, world, bs)
                sizes = [shard_bounds(n, r, world, bs) for r in range(world)]
                nb = [-(-(hi - lo) // bs) for lo, hi in sizes]
                assert max(nb) - min(nb) <= 1


def _enc():
    return synth.make_encodings(n_chars=40, n_words=50, n_upos=6, n_xpos=7, n_attrs=8, n_labels=9, num_langs=2)


class _FakeParser:
    """fills results as a pure function of the sentence and of its position in its batch (strings out of the
    encodings' tables, like the real one).  `reports` = hand the result arrays over through on_result, as
    cube_amd's Parser does; otherwise process_sharded reads them back from the words."""

    def __init__(self, reports=False):
        self._encodings = _enc()
        self.reports_results = reports

    def process(self, doc, collate, batch_size=4, num_workers=0, on_result=None):
        import numpy as np
        enc = self._encodings
        for start in range(0, len(doc.sentences), batch_size):
            batch = doc.sentences[start:start + batch_size]
            lmax = max(len(s.words) for s in batch)
            res = np.zeros((5, len(batch), lmax), dtype=np.int32)
            for b, s in enumerate(batch):
                for i, w in enumerate(s.words):
                    res[:, b, i] = ((i + lmax) % (len(s.words) + 1), lmax % len(enc.labels), len(w.word) % len(enc.upos_list),
                                    (i * 3) % len(enc.xpos_list), i % len(enc.attrs_list))
                    w.head = int(res[0, b, i])
                    w.label, w.upos = enc.labels[res[1, b, i]], enc.upos_list[res[2, b, i]]
                    w.xpos, w.attrs = enc.xpos_list[res[3, b, i]], enc.attrs_list[res[4, b, i]]
            if on_result is not None:
                on_result(batch, start, res)
        return doc


class _FakeLemmatizer:
    """a pure function of (word, UPOS, language), as Lemmatizer.process_bucketed is."""

    def process_bucketed(self, doc, collate, max_words=8192):
        for s in doc.sentences:
            for w in s.words:
                w.lemma = "%s|%s|%d" % (w.word.lower(), w.upos, s.lang_id)
        return doc


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import numpy as np
    from cube_amd.sharding import shard_result_ids, gather_result_ids
    enc = _enc()
    for tag, reports in (("", False), (".reports", True)):
        doc = synth.make_document(enc, [3 + (i * 7) % 11 for i in range(37)], seed=5, ext_word_emb=(4,))
        process_sharded(_FakeParser(reports), doc, None, batch_size=4, lemmatizer=_FakeLemmatizer())
        if rank == 0:
            with open(out + tag, "w", encoding="utf-8") as f:
                f.write(str(doc))
    # the arrays themselves: gathered parts, concatenated, equal the single-process result array
    doc = synth.make_document(enc, [3 + (i * 7) % 11 for i in range(37)], seed=5, ext_word_emb=(4,))
    bounds = [shard_bounds(len(doc.sentences), r, world, 4) for r in range(world)]
    lo, hi = bounds[rank]
    ids = shard_result_ids(_FakeParser(True), doc, None, lo, hi, 4)
    sizes = [sum(len(s.words) for s in doc.sentences[a:b]) for a, b in bounds]
    parts = gather_result_ids(ids, sizes, rank, world)
    if rank == 0:
        np.save(out + ".ids.npy", np.concatenate(parts))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_run_equals_single_process(tmp_path):
    out = str(tmp_path / "sharded.conllu")
    mp.spawn(_worker, args=(2, 29517, out), nprocs=2, join=True)
    import numpy as np
    from cube_amd.sharding import shard_result_ids
    enc = _enc()
    doc = synth.make_document(enc, [3 + (i * 7) % 11 for i in range(37)], seed=5, ext_word_emb=(4,))
    _FakeParser().process(doc, None, batch_size=4)
    _FakeLemmatizer().process_bucketed(doc, None)
    assert all(w.lemma != "_" for s in doc.sentences for w in s.words)
    assert open(out, encoding="utf-8").read() == str(doc)
    assert open(out + ".reports", encoding="utf-8").read() == str(doc)
    # the gathered int32 arrays of the two ranks = the result array of one process over the whole document
    single = shard_result_ids(_FakeParser(True), doc, None, 0, len(doc.sentences), 4)
    np.testing.assert_array_equal(np.load(out + ".ids.npy"), single)
    assert single.dtype == np.int32 and single.shape == (sum(len(s.words) for s in doc.sentences), 5)


def test_bucketed_batches_cover_every_sentence_within_the_token_budget():
    import numpy as np
    from cube_amd.sharding import bucketed_batches
    rng = np.random.default_rng(2)
    lens = np.clip(np.round(rng.normal(22, 10, size=3000)), 3, 80).astype(np.int64)
    groups = bucketed_batches(lens, max_tokens=8192)
    seen = np.concatenate(groups)
    assert sorted(seen.tolist()) == list(range(len(lens)))
    for g in groups:
        assert len(g) * int(lens[g].max()) <= 8192 and len(g) <= 1024
        assert (np.diff(lens[g]) >= 0).all()
    assert bucketed_batches([], 8192) == []
    assert [g.tolist() for g in bucketed_batches([9000, 2], 8192)] == [[1], [0]]  # an over-long sentence still gets a batch


def test_merged_reference_batches_keep_whole_batches_of_one_geometry():
    import numpy as np
    from cube_amd.sharding import merged_reference_batches
    rng = np.random.default_rng(4)
    keys = [(int(a), int(b)) for a, b in zip(rng.integers(3, 9, size=300), rng.integers(8, 11, size=300))]
    groups = merged_reference_batches(keys, 4, lambda k: 64 // k[0] * 4)
    assert sorted(i for g in groups for i in g) == list(range(300))          # every reference batch exactly once
    for g in groups:
        assert len({keys[i] for i in g}) == 1                                # one padding geometry per device batch
        assert g == sorted(g)                                                # reference order inside it
        assert len(g) * 4 <= max(4, 64 // keys[g[0]][0] * 4)                 # the cap, in items
    assert merged_reference_batches([], 4, 100) == []
    assert merged_reference_batches(["a", "b", "a"], 4, 1) == [[0], [2], [1]]  # cap below one batch: never split a batch
