"""cube_mst_decode_noalloc (include/cube_hip.h): the Chu-Liu-Edmonds decoder for callers that cannot
allocate — same results as cube_mst_decode, workspace in the library's built-in device array.

Collected last (file name): the round-end run is `pytest -x`.  Two calls in a row through the C ABI, the
second with a smaller input (the array keeps the first call's contents), each bit-equal to
cube_mst_decode on the same inputs; oversized inputs are refused with a clear error."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _probs(rng, B, L, sigma):
    logits = rng.standard_normal((B, L, L + 1)) * sigma
    e = np.exp(logits - logits.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(np.float32)


def _both(lib, hip, probs, lens, dev):
    from cube_amd.networks.utils import ChuLiuEdmondsDecoder
    B, L, _ = probs.shape
    p = torch.from_numpy(probs).to(dev)
    ln = torch.from_numpy(lens.astype(np.int32)).to(dev)
    ref_heads, ref_status = ChuLiuEdmondsDecoder(dev).decode_device(p, ln)   # cube_mst_decode, caller workspace
    heads = torch.full((B, L), -7, dtype=torch.int32, device=dev)
    status = torch.full((B,), -7, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        hip.check(lib.cube_mst_decode_noalloc(p.data_ptr(), ln.data_ptr(), B, L, heads.data_ptr(), status.data_ptr(),
                                              hip.stream_handle()), "cube_mst_decode_noalloc")
    torch.cuda.synchronize()
    return heads.cpu().numpy(), status.cpu().numpy(), ref_heads.cpu().numpy(), ref_status.cpu().numpy()


def test_mst_decode_noalloc_matches_mst_decode_on_two_calls_in_a_row():
    from cube_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    with torch.cuda.device(dev):
        hip.check(lib.cube_init(), "cube_init")
    rng = np.random.default_rng(99)
    # first call: 96 ragged sentences of up to 40 words, flat scores (many multi-root / no-root first trees, so
    # the task list, the candidate trees and their scores in the scratch array are all in use)
    B, L = 96, 40
    probs = _probs(rng, B, L, 1.0)
    lens = rng.integers(1, L + 1, size=B)
    lens[0] = L
    h, s, rh, rs = _both(lib, hip, probs, lens, dev)
    np.testing.assert_array_equal(s, rs)
    np.testing.assert_array_equal(h, rh)
    assert (rs == 0).all() and int((rh[0] == 0).sum()) == 1
    # second call, smaller: 5 sentences of up to 9 words on top of what the first call left in the array
    B2, L2 = 5, 9
    probs2 = _probs(rng, B2, L2, 0.5)
    lens2 = np.array([9, 1, 4, 9, 2])
    h2, s2, rh2, rs2 = _both(lib, hip, probs2, lens2, dev)
    np.testing.assert_array_equal(s2, rs2)
    np.testing.assert_array_equal(h2, rh2)
    # and against the NumPy restatement of the reference decoder (utils.py:565-577)
    from oracle import cle_oracle
    exp = cle_oracle.decode(probs2, lens2)
    for b in range(B2):
        np.testing.assert_array_equal(h2[b, :lens2[b]], exp[b])


def test_mst_decode_noalloc_refuses_what_does_not_fit():
    from cube_amd import hip
    lib = hip.load()
    dev = torch.device("cuda:0")
    B, L = 64, 128   # score matrices of the rooted re-runs live in the workspace from 96 nodes on: far beyond 4 MiB
    assert lib.cube_mst_workspace_bytes(B, L) > (4 << 20)
    p = torch.zeros((B, L, L + 1), dtype=torch.float32, device=dev)
    ln = torch.full((B,), L, dtype=torch.int32, device=dev)
    heads = torch.zeros((B, L), dtype=torch.int32, device=dev)
    status = torch.zeros((B,), dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.cube_mst_decode_noalloc(p.data_ptr(), ln.data_ptr(), B, L, heads.data_ptr(), status.data_ptr(),
                                         hip.stream_handle())
    assert rc == -1
    assert b"scratch" in lib.cube_last_error()
    torch.cuda.synchronize()
