"""Heads of two builds of cube_mst_decode on the same inputs (random, tie-heavy, zero-heavy): python scripts/mst_ab.py old.so [new.so]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cube_amd import hip  # noqa: E402

lib = hip.load()
libs = [ctypes.CDLL(sys.argv[1]), ctypes.CDLL(sys.argv[2]) if len(sys.argv) > 2 else lib]
for l in libs:
    l.cube_mst_decode.restype = ctypes.c_int
    l.cube_mst_decode.argtypes = hip.SIGNATURES["cube_mst_decode"][1]
dev = torch.device("cuda:0")
rng = np.random.default_rng(5)
st = torch.cuda.current_stream().cuda_stream


def run(l, probs, lens):
    B, L, _ = probs.shape
    heads = torch.zeros((B, L), dtype=torch.int32, device=dev)
    status = torch.zeros((B,), dtype=torch.int32, device=dev)
    nbytes = lib.cube_mst_workspace_bytes(B, L)
    ws = torch.zeros((nbytes + 7) // 8, dtype=torch.int64, device=dev)
    hip.check(l.cube_mst_decode(probs.data_ptr(), lens.data_ptr(), B, L, heads.data_ptr(), status.data_ptr(), ws.data_ptr(),
                                ws.numel() * 8, st), "cube_mst_decode")
    torch.cuda.synchronize()
    return heads.cpu().numpy(), status.cpu().numpy()


bad = 0
n = 0
for L in (5, 12, 33, 48, 64, 94, 100, 128, 129, 160, 254):
    for kind in ("soft0.05", "soft0.5", "soft3", "quant4", "quant16", "zeros0.5", "zeros0.8", "nan"):
        B = 24 if L <= 64 else 8
        if kind.startswith("soft"):
            probs = torch.softmax(torch.from_numpy(rng.standard_normal((B, L, L + 1)) * float(kind[4:])), -1).float()
        else:
            lv = 4 if kind == "quant4" else 16
            q = rng.integers(1, lv + 1, size=(B, L, L + 1)).astype(np.float32)
            if kind.startswith("zeros"):
                q[rng.random(q.shape) < float(kind[5:])] = 0.0
            if kind == "nan":
                q[rng.random(q.shape) < 0.01] = np.nan
            probs = torch.from_numpy(q / 64.0)
        lens = torch.from_numpy(np.array([L] + [int(rng.integers(1, L + 1)) for _ in range(B - 1)], dtype=np.int32))
        pd, ld = probs.to(dev).contiguous(), lens.to(dev)
        (h0, s0), (h1, s1) = run(libs[0], pd, ld), run(libs[1], pd, ld)
        same = np.array_equal(s0, s1) and all(np.array_equal(h0[b, :lens[b]], h1[b, :lens[b]]) for b in range(B) if s0[b] == 0)
        n += 1
        if not same:
            bad += 1
            print("DIFF L=%d %s: status %s vs %s, sentences that differ %s" % (
                L, kind, s0.tolist(), s1.tolist(), [b for b in range(B) if not np.array_equal(h0[b], h1[b])]))
print("%d of %d cases differ" % (bad, n))
