"""Wall-clock time of every collect() in a short bench run (the driver's form: --steps 20 --warmup 5): where a short timed
region loses against the steady state.   python scripts/step_times.py [steps] [warmup]"""
import os
import sys
import time
from collections import deque

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 5
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 4, 0, torch.device("cuda", 0), "")
bench.run_steps(p, batches, len(batches))
torch.cuda.synchronize()
for rep in range(3):
    bench.run_steps(p, batches, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stamps, subs = [], []
    inflight = deque()
    for i in range(steps):
        a = time.perf_counter()
        inflight.append(p.submit(batches[i % len(batches)]))
        subs.append(time.perf_counter() - a)
        if len(inflight) == p.N_SLOTS:
            p.collect(inflight.popleft())
            stamps.append(time.perf_counter() - t0)
    while inflight:
        p.collect(inflight.popleft())
        stamps.append(time.perf_counter() - t0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("run %d: %.3f ms per step; collect times (ms): %s" % (rep, dt / steps * 1e3, " ".join("%.2f" % (s * 1e3) for s in stamps)))
    print("        submit() host time (ms): %s" % " ".join("%.2f" % (s * 1e3) for s in subs))
