"""the pipeline `CubeObj.__call__` runs after tokenization — parser/tagger, then
the lemmatizer on the parser's UPOS tags — end to end from a Document (collate, upload, kernels,
write-back), on one MI355X, in the two throughput modes (`process_bucketed`).

    python scripts/bench_pipeline.py [--sentences 4096] [--lemma-steps 32]

Synthetic corpus of config 3 (lengths ~ clip(N(22,10), 3, 80)), random-init models.  A random
lemmatizer does not emit EOL at sensible places, so its greedy decode is capped at --lemma-steps
(the reference's only cap is 20 x S); everything else is what the product path does."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cube_amd import synth  # noqa: E402
from cube_amd.io_utils.config import LemmatizerConfig, ParserConfig  # noqa: E402
from cube_amd.networks.lemmatizer import Lemmatizer, Word2TargetCollate  # noqa: E402
from cube_amd.networks.parser import Parser  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sentences", type=int, default=4096)
    ap.add_argument("--lemma-steps", type=int, default=32)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    enc = synth.make_encodings()
    pcfg, lcfg = ParserConfig(), LemmatizerConfig()
    parser = Parser(pcfg, enc, language_codes=synth.EN_LANGUAGE_CODES, ext_word_emb=[300])
    parser.load_state_dict(synth.make_parser_state_dict(pcfg, enc, [300], seed=1234))
    parser.to("cuda:0")
    lem = Lemmatizer(lcfg, enc)
    lem.load_state_dict(synth.make_lemmatizer_state_dict(lcfg, enc))
    lem.to("cuda:0")
    lem.max_steps = a.lemma_steps
    lem.CHECK_EVERY = 1 << 30
    rng = np.random.default_rng(3000)
    lens = np.clip(np.round(rng.normal(22, 10, size=a.sentences)), 3, 80).astype(np.int64)
    doc = synth.make_document(enc, lens, seed=3000, ext_word_emb=(300,))
    pcol, lcol = MorphoCollate(enc), Word2TargetCollate(enc)
    n_words = int(lens.sum())
    tp, tl = [], []
    for rep in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        parser.process_bucketed(doc, pcol)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        lem.process_bucketed(doc, lcol)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if rep:  # the first pass sizes the workspaces
            tp.append(t1 - t0)
            tl.append(t2 - t1)
    keys = {(w.word, w.upos, s.lang_id) for s in doc.sentences for w in s.words}
    tp, tl = float(np.median(tp)), float(np.median(tl))
    print(json.dumps(dict(
        workload="config 4: %d sentences (%d words, %d distinct (word, UPOS, language)), parser.process_bucketed + "
                 "lemmatizer.process_bucketed, end to end from a Document" % (a.sentences, n_words, len(keys)),
        lemma_steps=a.lemma_steps, parser_s=round(tp, 4), lemmatizer_s=round(tl, 4),
        parser_sentences_per_s=round(a.sentences / tp, 1), lemmatizer_words_per_s=round(n_words / tl, 1),
        pipeline_sentences_per_s=round(a.sentences / (tp + tl), 1))))


if __name__ == "__main__":
    main()
