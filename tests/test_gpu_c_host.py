"""A host without Python at run time (examples/c_host.cpp, built by __graft_entry__.build()): packed-weights file ->
cube_packfile_* -> device memory the C program owns -> cube_gemm_bf16.  Its output must be, byte for byte, what the Python
host gets for the same layer and input through ctypes."""
import os
import subprocess

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C_HOST = os.path.join(ROOT, "nlp-cube_amd", "cube_amd", "lib", "c_host_example")


@pytest.mark.parametrize("prefix,M", [("char_pre_out", 300), ("pre_out", 77)])
def test_c_host_runs_a_packed_layer_like_the_python_host(tmp_path, prefix, M):
    from cube_amd import hip
    from cube_amd.networks import packing
    from cube_amd.networks.parser import Parser
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    if not os.path.exists(C_HOST):   # normally built beforehand by build(); a 3-second host compile if the tree came without it
        import __graft_entry__ as g
        g.build()
    assert os.path.exists(C_HOST), "examples/c_host.cpp is not built: python -c 'import __graft_entry__ as g; g.build()'"
    cfg, enc, sd = cases.parser_model(cases.FULL)
    p = Parser(cfg, enc, ext_word_emb=list(cases.FULL["ext"]))
    p.load_state_dict(sd)
    pack = str(tmp_path / "parser.cubepack")
    p.save_packed(pack)
    out = str(tmp_path / "out.bin")
    env = dict(os.environ)
    r = subprocess.run([C_HOST, pack, prefix, str(M), out], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    layer = p._pk_cpu[prefix]
    K, N = layer["K"], layer["N"]
    raw = np.fromfile(out, dtype=np.int16)
    assert raw.size == 2 * M * N
    c_hi, c_lo = raw[:M * N].reshape(M, N), raw[M * N:].reshape(M, N)
    # the same input, the same layer, through the Python host
    dev = torch.device("cuda:0")
    rr, cc = np.meshgrid(np.arange(M), np.arange(K), indexing="ij")
    x = torch.from_numpy((((rr * 131 + cc * 31 + 7) % 2001).astype(np.float32) / np.float32(1000.0) - np.float32(1.0)))
    A = hip.Pair.split(x).to(dev)
    W = packing.to_device(layer["W"], dev)
    o = hip.Pair(torch.zeros((M, N), dtype=torch.bfloat16, device=dev), torch.zeros((M, N), dtype=torch.bfloat16, device=dev))
    hip.gemm(A, K, M, K, W, N, layer["bias"].to(dev), hip.EPI_BF16_TANH, out0=o, ldo0=N, n_valid=N)
    torch.cuda.synchronize()
    assert np.array_equal(o.hi.cpu().view(torch.int16).numpy(), c_hi)
    assert np.array_equal(o.lo.cpu().view(torch.int16).numpy(), c_lo)
    assert np.abs(c_hi).max() > 0
