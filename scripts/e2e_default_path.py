"""What `Cube.__call__` costs with the reference's own defaults (Parser.process batch_size=4 sentences,
Lemmatizer.process batch_size=4 words), end to end from a Document on one MI355X, with and without
merging reference batches of equal padding geometry into device batches (results are identical by
construction: `sharding.merged_reference_batches`, GPU tests `test_merged_reference_batches_*`).

    python scripts/e2e_default_path.py [--sentences 4096]

Ragged synthetic corpus of SURVEY §8d config 3, random-init models; the lemmatizer's greedy decode is
capped at 32 steps (a random model never emits EOL)."""
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


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sentences", type=int, default=4096)
    ap.add_argument("--unmerged-sentences", type=int, default=512)
    a = ap.parse_args()
    enc = synth.make_encodings()
    pcfg, lcfg = ParserConfig(), LemmatizerConfig()
    parser = Parser(pcfg, enc, language_codes=synth.EN_LANGUAGE_CODES, ext_word_emb=[300])
    parser.load_state_dict(synth.make_parser_state_dict(pcfg, enc, [300], seed=1234))
    parser.to("cuda:0")
    lem = Lemmatizer(lcfg, enc)
    lem.load_state_dict(synth.make_lemmatizer_state_dict(lcfg, enc))
    lem.to("cuda:0")
    lem.max_steps, lem.CHECK_EVERY = 32, 1 << 30
    rng = np.random.default_rng(3000)
    lens = np.clip(np.round(rng.normal(22, 10, size=a.sentences)), 3, 80).astype(np.int64)
    doc = synth.make_document(enc, lens, seed=3000, ext_word_emb=(300,))
    small = synth.make_document(enc, lens[:a.unmerged_sentences], seed=3000, ext_word_emb=(300,))
    pcol, lcol = MorphoCollate(enc), Word2TargetCollate(enc)
    out = {"workload": "%d ragged sentences (%d words), reference defaults: parser batch_size=4 sentences, lemmatizer "
                       "batch_size=4 words" % (a.sentences, int(lens.sum()))}
    for name, d, merge in (("merged", doc, True), ("unmerged", small, False)):
        parser.process(d, pcol, merge=merge)        # sizes the workspaces
        tp = timed(lambda: parser.process(d, pcol, merge=merge))
        nw = sum(len(s.words) for s in d.sentences)
        if merge:
            lem.process(d, lcol, merge=True)
            tl = timed(lambda: lem.process(d, lcol, merge=True))
        else:   # 4 words per launch sequence: time a slice of the words only
            from cube_amd.sharding import _Slice
            part = _Slice(d, 0, 64)
            nw = sum(len(s.words) for s in part.sentences)
            lem.process(part, lcol, merge=False)
            tl = timed(lambda: lem.process(part, lcol, merge=False))
        out[name] = dict(parser_sentences_per_s=round(len(d.sentences) / tp, 1), lemmatizer_words_per_s=round(nw / tl, 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
