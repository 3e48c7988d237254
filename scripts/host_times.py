"""Host-side timing of the pipelined step: how long `submit` takes to enqueue a batch and how long
`collect` waits (bench workload, config 1).  python scripts/host_times.py [steps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 4, 0, dev, "")
bench.run_steps(p, batches, 8)
torch.cuda.synchronize()
sub, col = [], []
pending = None
t00 = time.perf_counter()
for i in range(n):
    t0 = time.perf_counter()
    t = p.submit(batches[i % 4])
    t1 = time.perf_counter()
    if pending is not None:
        p.collect(pending)
    t2 = time.perf_counter()
    pending = t
    sub.append(t1 - t0)
    col.append(t2 - t1)
p.collect(pending)
tot = time.perf_counter() - t00
print("step %.3f ms; submit host %.3f ms (min %.3f); collect wait %.3f ms" %
      (tot / n * 1e3, np.mean(sub) * 1e3, np.min(sub) * 1e3, np.mean(col) * 1e3))
