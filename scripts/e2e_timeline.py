"""Host timeline of Parser.process: every call of the host stages with its thread, start and end, for a few steady-state
batches.    python scripts/e2e_timeline.py [batch_size] [first batch] [batches]"""
import copy
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from cube_amd import synth  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
first = int(sys.argv[2]) if len(sys.argv) > 2 else 8
count = int(sys.argv[3]) if len(sys.argv) > 3 else 4
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(1, 1, 0, dev, "")
col = MorphoCollate(enc)
big = synth.make_document(enc, [32] * 1024, seed=77, ext_word_emb=(300,))
k = 0
while len(big.sentences) < 16384:
    big.sentences.extend(synth.make_document(enc, [32] * 1024, seed=1078 + k, ext_word_emb=(300,)).sentences)
    k += 1
log = []


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            log.append((threading.current_thread().name[:12], label, t, time.perf_counter()))
    setattr(obj, name, w)


from cube_amd import host as _host  # noqa: E402
hm = _host.load()
timed(hm, "featurize", "C featurize")
timed(hm, "write_back", "C write_back")
for name in ("_host_batch", "_upload", "submit", "collect", "_write_back"):
    timed(p, name)
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


for name in ("_upload", "_front", "_back", "_decode"):
    dev_timed(name)
p.process(big, col, batch_size=bs)
log.clear()
dev_log.clear()
torch.cuda.synchronize()
base = torch.cuda.Event(enable_timing=True)
base.record()
torch.cuda.synchronize()
slow = []
if os.environ.get("SLOW_CALLS"):
    stack = []

    def prof(frame, event, arg):
        if event in ("call", "c_call"):
            stack.append(time.perf_counter())
        elif event in ("return", "c_return", "c_exception"):
            if stack:
                t = stack.pop()
                d = time.perf_counter() - t
                if d > 0.3e-3:
                    name = getattr(arg, "__qualname__", None) or getattr(arg, "__name__", None) or repr(arg)[:60] if event != "return" else frame.f_code.co_name
                    slow.append((t, d, event, name))
    sys.setprofile(prof)
t0 = time.perf_counter()
p.process(big, col, batch_size=bs)
dt = time.perf_counter() - t0
sys.setprofile(None)
print("process: %.1f ms for %d sentences = %.0f sentences/s, feature_thread=%s" % (dt * 1e3, len(big.sentences), len(big.sentences) / dt, p.feature_thread))
subs = [e for e in log if e[1] == "submit"]
print("submit calls at (ms since the start of process):", " ".join("%.1f" % ((e[2] - t0) * 1e3) for e in subs), "| end %.1f" % (dt * 1e3))
for nm in ("_host_batch", "_upload", "submit", "collect", "_write_back"):
    v = [e[3] - e[2] for e in log if e[1] == nm]
    print("  %-12s %3d calls, %.2f ms total, first %.2f, last %.2f" % (nm, len(v), sum(v) * 1e3, v[0] * 1e3, v[-1] * 1e3))
ta = subs[first][2]
tb = subs[min(first + count, len(subs) - 1)][2]
for th, name, s, e in sorted(log, key=lambda r: r[2]):
    if ta <= s < tb:
        print("%9.3f .. %9.3f  (%6.3f ms)  %-12s %s" % ((s - ta) * 1e3, (e - ta) * 1e3, (e - s) * 1e3, th, name))

torch.cuda.synchronize()
print("device stages (ms since the start of process; host time of the enqueue in brackets):")
for name, th, e0, e1 in dev_log:
    a, b = base.elapsed_time(e0), base.elapsed_time(e1)
    if (ta - t0) * 1e3 - 6 <= a < (tb - t0) * 1e3 + 6:
        print("%9.3f .. %9.3f  (%6.3f ms)  %-8s [enqueued by %9.3f]" % (a - (ta - t0) * 1e3, b - (ta - t0) * 1e3, b - a, name, (th - ta) * 1e3))

for t, d, ev, name in slow:
    if ev != "return" and ta <= t < tb + 10e-3:
        print("slow C call at %9.3f: %6.3f ms  %s" % ((t - ta) * 1e3, d * 1e3, name))
