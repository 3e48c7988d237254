"""How long the host needs to enqueue one batch (GPU box)."""
import sys, os, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
import bench
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 2, 0, torch.device("cuda:0"), "")
bench.run_steps(p, batches, 3)
torch.cuda.synchronize()
ts = []
for i in range(10):
    t0 = time.perf_counter()
    r = p.decode_batch_prepared(batches[i % 2])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((t1 - t0, t2 - t0))
print("enqueue ms %.2f  total ms %.2f" % (1e3 * sum(a for a, b in ts) / 10, 1e3 * sum(b for a, b in ts) / 10))
