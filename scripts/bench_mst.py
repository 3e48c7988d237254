"""Times cube_mst_decode alone (HIP events) on softmax-of-noise arc probabilities, the shape random-init weights give.
    python scripts/bench_mst.py [B] [L] [sigma] [lib.so]    (lib.so: an A/B build of csrc/mst.hip + csrc/rowops.hip)
With a profile build (cube_mst_prof exported) the per-phase cycle counts summed over all tasks are printed too."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cube_amd import hip  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = int(sys.argv[2]) if len(sys.argv) > 2 else 128
sigma = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
lib = hip.load()
alt = ctypes.CDLL(sys.argv[4]) if len(sys.argv) > 4 else lib
fn = alt.cube_mst_decode
fn.restype = ctypes.c_int
fn.argtypes = hip.SIGNATURES["cube_mst_decode"][1]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
if os.environ.get("MST_BENCH_MODEL") == "1":
    # the arc probabilities of the bench's own first batch (random-init full-size model): what bench.py's decode stage sees
    import bench
    cfg_id = {(256, 32): 1, (64, 128): 2}[(B, L)]
    p_, *_rest, batches, _docs = bench.build_model_and_batches(cfg_id, 1, 0, dev, "")
    r_ = p_.decode_batch_prepared(batches[0])
    probs = r_["arc_probs"].clone().contiguous()
    del p_, r_
else:
    probs = torch.softmax(torch.randn((B, L, L + 1), generator=g) * sigma, -1).to(dev).contiguous()
lens = torch.full((B,), L, dtype=torch.int32, device=dev)
heads = torch.empty((B, L), dtype=torch.int32, device=dev)
status = torch.empty((B,), dtype=torch.int32, device=dev)
nbytes = lib.cube_mst_workspace_bytes(B, L)
ws = torch.empty((nbytes + 7) // 8, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream


def run():
    hip.check(fn(probs.data_ptr(), lens.data_ptr(), B, L, heads.data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel() * 8, st),
              "cube_mst_decode")


for _ in range(2):
    run()
torch.cuda.synchronize()
ntasks = int(ws[:1].view(torch.int32)[0].item())
reps = 5
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(reps):
    run()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
h = heads.cpu().numpy()
print("B=%d L=%d sigma=%g: %.3f ms per batch, %d rooted tasks, %.0f sentences/s, heads checksum %d, status!=0: %d"
      % (B, L, sigma, ms, ntasks, B / ms * 1e3, int((h.astype(np.int64) * (1 + np.arange(L))).sum()), int((status != 0).sum().item())))
if hasattr(alt, "cube_mst_prof"):
    out = (ctypes.c_longlong * 16)()
    alt.cube_mst_prof(out)
    v = np.array(list(out), dtype=np.float64)
    names = ["scan", "cycles", "lists", "rec/product", "entry/exit", "update", "expand"]
    tot = v[10]
    print("tasks %d, levels/task %.1f, cycles/task %.0f" % (v[9], v[8] / v[9], tot / v[9]))
    for i, nme in enumerate(names):
        print("  %-12s %5.1f %%  %.0f cycles/level" % (nme, 100 * v[i] / tot, v[i] / max(v[8], 1)))
