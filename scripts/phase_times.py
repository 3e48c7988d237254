"""Front-only / back-only / serial / pipelined step times of the parser (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
import torch
import bench
dev = torch.device("cuda:0")
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(int(os.environ.get("CONFIG", "1")), 4, 0, dev, "")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
fr = [None]
def front(i=0): fr[0] = p._front(batches[i % 4], 0)
def back(i=0): p._decode(batches[i % 4], p._back(batches[i % 4], dict(fr[0]), 0), 0)
def serial(i=0): p.decode_batch_prepared(batches[i % 4])
front(); torch.cuda.synchronize()
print("front only  %.3f ms" % timeit(front))
print("back only   %.3f ms" % timeit(back))
print("serial      %.3f ms" % timeit(serial))
t0 = time.perf_counter(); bench.run_steps(p, batches, 3); torch.cuda.synchronize()
t0 = time.perf_counter(); bench.run_steps(p, batches, 40); torch.cuda.synchronize()
print("pipelined   %.3f ms" % ((time.perf_counter() - t0) / 40 * 1e3))
