"""Device timeline of the three stages (HIP events around _front / _back / _decode on their streams) for the pre-collated
batches of one batch size, pipelined as bench.py runs them.    python scripts/stage_timeline.py B [steps shown]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import torch  # noqa: E402
import bench  # noqa: E402

B = int(sys.argv[1])
shown = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda", 0)
bench.CONFIGS[1] = dict(B=B, L=32, name="sweep")
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 4, 0, dev, "")
dev_log = []


def dev_timed(name):
    fn = getattr(p, name)

    def w(*a, **k):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn(*a, **k)
        e1.record()
        dev_log.append((name, time.perf_counter(), e0, e1))
        return r
    setattr(p, name, w)


bench.run_steps(p, batches, 30)
torch.cuda.synchronize()
for name in ("_front", "_back", "_decode"):
    dev_timed(name)
base = torch.cuda.Event(enable_timing=True)
base.record()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 24
bench.run_steps(p, batches, n)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("B = %d: %.3f ms per batch, %.0f sentences/s" % (B, dt / n * 1e3, n * B / dt))
k = 0
for name, th, e0, e1 in dev_log:
    a, b = base.elapsed_time(e0), base.elapsed_time(e1)
    if name == "_front":
        k += 1
    if 10 <= k < 10 + shown:
        print("%9.3f .. %9.3f  (%6.3f ms)  %-8s [enqueued at %9.3f]" % (a, b, b - a, name, (th - t0) * 1e3))
