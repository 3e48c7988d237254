"""cProfile of Parser.process at batch 256 (GPU box): where the host time of the end-to-end path goes."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
import torch
import bench
from cube_amd import synth
from cube_amd.networks.utils import MorphoCollate
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 1, 0, torch.device("cuda:0"), "")
col = MorphoCollate(enc)
doc = synth.make_document(enc, [32] * 4096, seed=77, ext_word_emb=(300,))
p.process(doc, col, batch_size=256)
pr = cProfile.Profile(); pr.enable(); p.process(doc, col, batch_size=256); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
