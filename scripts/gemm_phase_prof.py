"""Per-workgroup phase times of the int8x2 char-conv tile (needs the PROF build of scripts/gemm_exp_build.py in place of
the library): the bench batches are decoded serially, then the stamps of the last launch (char layer 4) are read."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
from cube_amd import hip  # noqa: E402

p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 2, 0, torch.device("cuda", 0), "")
for _ in range(3):
    for bt in batches:
        p.decode_batch_prepared(bt)
torch.cuda.synchronize()
lib = hip.load()
out = (ctypes.c_longlong * (4 * 4096))()
assert lib.cube_gemm_prof(out) == 0
v = np.array(list(out), dtype=np.int64).reshape(4096, 4)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 736
v = v[:nb]
v = v[v[:, 0] > 0]
t0 = v[:, 0].min()
us = (v - t0) / 100.0
pro, loop, epi = us[:, 1] - us[:, 0], us[:, 2] - us[:, 1], us[:, 3] - us[:, 2]
print("blocks %d: prologue %.2f us, loop %.2f us, epilogue %.2f us (means); launch span %.1f us" % (len(v), pro.mean(), loop.mean(), epi.mean(), us[:, 3].max()))
order = np.argsort(us[:, 0])
for q in (0, 100, 255, 256, 300, 511, 512, 600, len(v) - 1):
    if q < len(v):
        i = order[q]
        print("  start rank %4d (block %4d): start %.1f  prologue %.2f  loop %.2f  epilogue %.2f  end %.1f" % (q, i, us[i, 0], pro[i], loop[i], epi[i], us[i, 3]))
