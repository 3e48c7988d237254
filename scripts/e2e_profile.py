"""Where the host time of Parser.process goes: wall time inside each stage of `_run_batches`, main thread and
featurisation worker.    python scripts/e2e_profile.py [batch_size]"""
import copy
import os
import sys
import time
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from cube_amd import synth  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 1, 0, dev, "")
col = MorphoCollate(enc)
doc = synth.make_document(enc, [32] * 1024, seed=77, ext_word_emb=(300,))
big = copy.deepcopy(doc)
k = 0
while len(big.sentences) < 16384:  # distinct sentences: copies would let Parser.dedup_words skip most of the character CNN
    big.sentences.extend(synth.make_document(enc, [32] * 1024, seed=1078 + k, ext_word_emb=(300,)).sentences)
    k += 1
acc = defaultdict(float)
cnt = defaultdict(int)


def timed(name):
    fn = getattr(p, name)

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] += time.perf_counter() - t
            cnt[name] += 1
    setattr(p, name, w)


from cube_amd import host as _host  # noqa: E402
_hm = _host.load()


def timed_obj(obj, name, label):
    fn = getattr(obj, name)

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] += time.perf_counter() - t
            cnt[label] += 1
    setattr(obj, name, w)


timed_obj(_hm, "featurize", "C featurize")
timed_obj(_hm, "write_back", "C write_back")
timed_obj(col, "cp_table32", "cp_table32")
for name in ("_staging_acquire", "_host_batch", "_upload", "submit", "collect", "_write_back", "_front", "_back", "_decode"):
    timed(name)
p.process(big, col, batch_size=bs)
acc.clear()
cnt.clear()
from cube_amd import host  # noqa: E402
host.load().timing()
torch.cuda.synchronize()
t0 = time.perf_counter()
p.process(big, col, batch_size=bs)
dt = time.perf_counter() - t0
print("process: %.1f ms for %d sentences = %.0f sentences/s" % (dt * 1e3, len(big.sentences), len(big.sentences) / dt))
for k in acc:
    print("  %-12s %4d calls  %7.2f ms total  %6.3f ms each" % (k, cnt[k], acc[k] * 1e3, acc[k] / cnt[k] * 1e3))
c, t0_, t1_, t2_, t3_, *sub = host.load().timing()
print("featurize (C), ms per call: sizes %.3f, objects %.3f, index arrays %.3f, embedding copies %.3f (%d calls)"
      % (t0_ / c * 1e3, t1_ / c * 1e3, t2_ / c * 1e3, t3_ / c * 1e3, c))
print("  of the index arrays: code points %.3f, repeated words %.3f, length groups %.3f, character rows %.3f"
      % tuple(x / c * 1e3 for x in sub[:4]))
if len(sub) >= 7:
    print("  of the objects: Word objects (sweeps A, B) %.3f, strings + word ids (C) %.3f, embedding lists and arrays (D, E) %.3f"
          % tuple(x / c * 1e3 for x in sub[4:7]))
