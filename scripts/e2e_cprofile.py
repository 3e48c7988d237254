"""cProfile of Parser.process(Document) on the bench's config-1 model: which Python functions the host time of the
end-to-end path goes to.    python scripts/e2e_cprofile.py [batch_size] [n_sentences]"""
import copy
import cProfile
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from cube_amd import synth  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 1, 0, dev, "")
col = MorphoCollate(enc)
doc = synth.make_document(enc, [32] * 1024, seed=77, ext_word_emb=(300,))
big = copy.deepcopy(doc)
k = 0
while len(big.sentences) < ns:  # distinct sentences: copies would let Parser.dedup_words skip most of the character CNN
    big.sentences.extend(synth.make_document(enc, [32] * 1024, seed=1078 + k, ext_word_emb=(300,)).sentences)
    k += 1
p.process(big, col, batch_size=bs, num_workers=0)  # warm-up: allocations, staging ring
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
p.process(big, col, batch_size=bs, num_workers=0)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
