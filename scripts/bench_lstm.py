"""Times cube_bilstm_recurrent_i8 / cube_bilstm_recurrent alone (HIP events, 20 launches): step latency x steps vs wall.
    python scripts/bench_lstm.py [B] [T] [H] [lib.so]     (lib.so: an A/B build of csrc/lstm_i8.hip alone)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from cube_amd import hip  # noqa: E402
from cube_amd.networks import packing  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 33
H = int(sys.argv[3]) if len(sys.argv) > 3 else 200
lib = hip.load()
alt = ctypes.CDLL(sys.argv[4]) if len(sys.argv) > 4 else lib
fn8 = alt.cube_bilstm_recurrent_i8
fn8.restype = ctypes.c_int
fn8.argtypes = hip.SIGNATURES["cube_bilstm_recurrent_i8"][1]
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
rows1 = packing.lstm_gate_rows(H)
G = len(rows1)
v = np.where(rows1 >= 0)[0]
whh = []
for d in range(2):
    w = torch.from_numpy(rng.uniform(-1, 1, size=(4 * H, H)) / np.sqrt(H)).float()
    wp = torch.zeros((G, packing.p32(H)), dtype=torch.float32)
    wp[torch.from_numpy(v), :H] = w[torch.from_numpy(rows1[v])]
    whh.append(wp)
pk = packing.pack_lstm_i8(whh, H, int(os.environ["WLO"]) if "WLO" in os.environ else None)
gx = torch.randn((B * T, 2 * G), device=dev)
ld = packing.p32(2 * H + 64)
out = torch.zeros((B * T, ld), dtype=torch.bfloat16, device=dev)
out_lo = torch.zeros_like(out)
d_hi, d_lo, d_sc = pk["hi"].to(dev), pk["lo"].to(dev), pk["scale"].to(dev)
st = torch.cuda.current_stream().cuda_stream


def run8():
    hip.check(fn8(gx.data_ptr(), d_hi.data_ptr(), d_lo.data_ptr(), d_sc.data_ptr(), B, T, H, pk["lo_bits"], out.data_ptr(), out_lo.data_ptr(),
                  ld, 0, st), "cube_bilstm_recurrent_i8")


# bf16x3 kernel: W_hh fragments streamed from L2
U, KS = G // 64, packing.p32(H) // 32
w4 = torch.stack(whh, 0).view(2, U, 4, 16, KS, 4, 8)
frag = w4.permute(0, 1, 2, 4, 5, 3, 6).reshape(2, U, 4, KS, 64, 8).contiguous()
fr = hip.Pair.split(frag).to(dev)


def run16():
    hip.check(lib.cube_bilstm_recurrent(gx.data_ptr(), fr.hi.data_ptr(), fr.lo.data_ptr(), B, T, H, out.data_ptr(),
                                        out_lo.data_ptr(), ld, 0, st), "cube_bilstm_recurrent")


for name, fn in (("int8x2 resident (cube_bilstm_recurrent_i8)", run8), ("bf16x3 streamed (cube_bilstm_recurrent)", run16)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        fn()
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print("%-45s B=%d T=%d H=%d: %.1f us per layer, %.2f us per step" % (name, B, T, H, us, us / T))
