import ctypes, os, sys
ROOT = "/root/repo"
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np, torch, bench
from cube_amd import hip
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 2, 0, torch.device("cuda", 0), "")
for _ in range(3):
    for bt in batches: p.decode_batch_prepared(bt)
torch.cuda.synchronize()
out = (ctypes.c_longlong * (8 * 4096))()
assert hip.load().cube_gemm_prof(out) == 0
v = np.array(list(out), dtype=np.int64).reshape(4096, 8)[:736, :5]
v = v[v[:, 0] > 0]
d = np.diff(v, axis=1) / 100.0
print("blocks", len(v), "entry->before DMA issue %.2f us, DMA issue %.2f, row tables %.2f, barrier wait %.2f" % tuple(d.mean(0)))
