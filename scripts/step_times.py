"""Completion time of every step of the driver's form (W warm-up steps, synchronise, K timed steps, synchronise):
    python scripts/step_times.py [K] [W]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 1, 0, dev, "")
for bt in batches:
    p.collect(p.submit(bt))
bench.run_steps(p, batches, W)
torch.cuda.synchronize()
for rep in range(3):
    from collections import deque
    inflight, done = deque(), []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        inflight.append(p.submit(batches[i % len(batches)]))
        if len(inflight) >= p.in_flight_for(batches[0]["T"]):
            p.collect(inflight.popleft())
            done.append(time.perf_counter() - t0)
    while inflight:
        p.collect(inflight.popleft())
        done.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    gaps = [done[0]] + [b - a for a, b in zip(done, done[1:])]
    print("rep %d: total %.2f ms = %.0f sentences/s; completion gaps (ms): %s" % (rep, total * 1e3, K * batches[0]["B"] / total, " ".join("%.2f" % (g * 1e3) for g in gaps)))
