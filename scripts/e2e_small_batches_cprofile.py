"""cProfile of Parser.process on a RAGGED document at the reference's default batch_size = 4 (reference batches of different
geometry cannot be merged: 64 small device batches per 256 sentences): where the host time of a small batch goes.
    python scripts/e2e_small_batches_cprofile.py"""
import cProfile
import os
import pstats
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
lens = np.clip(np.round(rng.normal(22, 10, size=1024)), 3, 80).astype(np.int64)
doc = synth.make_document(enc, lens, seed=5, ext_word_emb=(300,))
for _ in range(2):
    p.process(doc, col)
torch.cuda.synchronize()
t0 = time.perf_counter()
p.process(doc, col)
print("1024 ragged sentences at batch_size 4: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
p.process(doc, col)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(30)
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
