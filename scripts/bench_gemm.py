"""Micro-benchmark of cube_gemm_bf16 at the char-conv shape (GPU box): python scripts/bench_gemm.py [split] [M] [res]"""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
from cube_amd import hip
split = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
W_ = int(os.environ.get('SEQ', '14'))
M = int(sys.argv[2]) if len(sys.argv) > 2 else 8192 * W_
Kc, half, taps = int(os.environ.get('KC', '320')), 256, 5
use_tab = os.environ.get('TAB', '0') == '1'
N = 2 * half
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
A = hip.Pair.split(torch.randn((M, Kc), generator=g), split).to(dev)
Wt = hip.Pair.split(torch.randn((N, taps * Kc), generator=g) / 40, split).to(dev)
bias = torch.randn(N, generator=g).to(dev)
use_res = (sys.argv[3] if len(sys.argv) > 3 else "0") == "1"
res = torch.zeros((M, half), device=dev) if use_res else None
out = hip.Pair(torch.zeros((M, max(Kc, half)), dtype=torch.bfloat16, device=dev), torch.zeros((M, max(Kc, half)), dtype=torch.bfloat16, device=dev) if split else None)
i8out = os.environ.get('I8OUT', '0') == '1'
tab = torch.randn((7, 32, N), generator=g).to(dev) if use_tab else None
grp = torch.randint(0, 7, (M,), generator=g).to(torch.int32).to(dev) if use_tab else None
def run():
    hip.gemm(A, Kc, M, Kc, Wt, N, bias, hip.EPI_GATE, out0=out, ldo0=max(Kc, half), col0=0, out0_int8=i8out, taps=taps, seq_len=W_, n_valid=half,
             res=res, ld_res=half if use_res else 0, res_mode=1 if use_res else 0, bias_tab=tab, row_group=grp)
for _ in range(3): run()
torch.cuda.synchronize()
n = 20
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
fl = 2.0 * M * taps * Kc * N
print("split=%d M=%d: %.3f ms  algorithmic %.1f TFLOP/s  executed-MFMA %.1f TFLOP/s" % (split, M, ms, fl / ms / 1e9, fl * (3 if split else 1) / ms / 1e9))
