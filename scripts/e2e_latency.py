"""Latency of Parser.process(Document) for small documents (a serving call: one request = one short text), one MI355X:
    python scripts/e2e_latency.py      -> median / p90 milliseconds for documents of 1, 4, 16, 64, 256 sentences of ~22 words"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tests import cases  # noqa: E402
from cube_amd import synth  # noqa: E402
from cube_amd.networks.parser import Parser  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402

cfg, enc, sd = cases.parser_model(cases.FULL)
p = Parser(cfg, enc, ext_word_emb=[300])
p.load_state_dict(sd)
p.to("cuda:0")
col = MorphoCollate(enc)
rng = np.random.default_rng(9)
out = {}
for n in (1, 4, 16, 64, 256):
    docs = []
    for k in range(40):
        lens = np.clip(np.round(rng.normal(22, 10, size=n)), 3, 80).astype(np.int64)
        docs.append(synth.make_document(enc, lens, seed=100 + k, ext_word_emb=(300,)))
    for d in docs[:10]:
        p.process(d, col)
    torch.cuda.synchronize()
    ts = []
    for d in docs[10:]:
        t0 = time.perf_counter()
        p.process(d, col)
        ts.append((time.perf_counter() - t0) * 1e3)
    out["%d_sentences" % n] = dict(median_ms=round(float(np.median(ts)), 3), p90_ms=round(float(np.percentile(ts, 90)), 3),
                                   sentences_per_s=round(n / (np.median(ts) * 1e-3)))
# the same 256-sentence documents with larger reference batches, and length-bucketed (no counterpart in the reference)
docs = []
for k in range(30):
    lens = np.clip(np.round(rng.normal(22, 10, size=256)), 3, 80).astype(np.int64)
    docs.append(synth.make_document(enc, lens, seed=500 + k, ext_word_emb=(300,)))
for name, fn in (("256_sentences_batch_size_32", lambda d: p.process(d, col, batch_size=32)),
                 ("256_sentences_batch_size_256", lambda d: p.process(d, col, batch_size=256)),
                 ("256_sentences_bucketed", lambda d: p.process_bucketed(d, col))):
    for d in docs[:8]:
        fn(d)
    torch.cuda.synchronize()
    ts = []
    for d in docs[8:]:
        t0 = time.perf_counter()
        fn(d)
        ts.append((time.perf_counter() - t0) * 1e3)
    out[name] = dict(median_ms=round(float(np.median(ts)), 3), p90_ms=round(float(np.percentile(ts, 90)), 3),
                     sentences_per_s=round(256 / (np.median(ts) * 1e-3)))
print(json.dumps(dict(workload="Parser.process(Document), ragged sentences of ~22 words, default batch_size", **out)))
