"""GPU: sub-word -> word pooling of the external-embedding stage (cube_subword_pool, csrc/pool.hip; LMHelperHF of
cube_amd/networks/lm.py) against the reference's own outputs (golden G12) and the oracle -- bit-equal."""
import ctypes
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def test_lm_helper_apply_matches_reference(dev, golden_dir):
    """LMHelperHF.apply (per-sentence model calls, 512-piece windows, pooling on the device) fills word.emb with exactly the
    13 x 768 values the reference's LMHelperHF.apply produced from the same (synthetic) tokenizer and hidden states."""
    from cube_amd.networks.lm import LMHelperHF
    g = np.load(os.path.join(golden_dir, "g12_lm_pooling.npz"))
    _, enc, _ = cases.parser_model(cases.FULL)
    doc = cases.lm_document(enc)
    h = LMHelperHF(device=dev, splitter=cases.FakeSplitter(), encoder=cases.FakeEncoder())
    assert h.get_embedding_size() == [cases.LM_WIDTH] * cases.LM_LAYERS
    h.apply(doc)
    for i, s in enumerate(doc.sentences):
        a = np.stack([np.stack(w.emb) for w in s.words])
        assert a.dtype == np.float32 and all(v.flags["C_CONTIGUOUS"] for w in s.words for v in w.emb)
        np.testing.assert_array_equal(a if i < 3 else a[:, :, ::16], g["sent%d" % i])
    # `_compute_we` keeps the reference's return format: one K*E vector per word, and batches of several sentences work
    we = h._compute_we(doc.sentences[:3])
    assert len(we) == 18 and we[0].shape == (cases.LM_LAYERS * cases.LM_WIDTH,)


@pytest.mark.parametrize("S,P,E,K", [(3, 40, 768, 13), (1, 700, 64, 2), (5, 9, 4, 1), (2, 64, 1024, 32)])
def test_subword_pool_kernel_matches_oracle(dev, S, P, E, K):
    """cube_subword_pool through the C ABI on random hidden states and random spans (0 .. 9 pieces, strided layers):
    bit-equal to the NumPy restatement of lm.py:183-197."""
    from cube_amd import hip
    from oracle import lm_oracle
    lib = hip.load()
    rng = np.random.default_rng(S * 100 + P)
    pad = 8   # a layer whose piece stride is wider than E
    layers = [torch.from_numpy(rng.standard_normal((S, P, E + pad)).astype(np.float32)).to(dev) for _ in range(K)]
    spans, w2p = [], {}

    class Sent:
        def __init__(self):
            self.words = []
    sents = []
    for s in range(S):
        sent, pos = Sent(), 1
        while pos < P - 1:
            n = int(rng.integers(0, 10))
            n = min(n, P - 1 - pos)
            w2p[(s, len(sent.words))] = [(s, pos + j) for j in range(n)]
            spans.append((s, pos, n))
            sent.words.append(None)
            pos += max(n, 1) if n else 0
            if n == 0 and rng.random() < 0.5:
                pos += 1
        sents.append(sent)
    sp = torch.from_numpy(np.ascontiguousarray(np.asarray(spans, dtype=np.int32).T)).to(dev)
    n_words = len(spans)
    out = torch.full((K, n_words, E), -7.0, dtype=torch.float32, device=dev)
    ptrs = (ctypes.c_void_p * K)(*[t.data_ptr() for t in layers])
    hip.check(lib.cube_subword_pool(ptrs, K, P * (E + pad), E + pad, E, sp[0].data_ptr(), sp[1].data_ptr(), sp[2].data_ptr(),
                                    n_words, out.data_ptr(), hip.stream_handle()), "cube_subword_pool")
    torch.cuda.synchronize()
    we = np.concatenate([t.cpu().numpy()[:, :, :E] for t in layers], axis=-1)
    exp = np.stack(lm_oracle.pool(we, sents, w2p)).reshape(n_words, K, E).transpose(1, 0, 2)
    np.testing.assert_array_equal(out.cpu().numpy(), exp)
    # bad arguments are refused, not launched
    assert lib.cube_subword_pool(ptrs, K, P * (E + pad), E + pad, E + 1, sp[0].data_ptr(), sp[1].data_ptr(), sp[2].data_ptr(),
                                 n_words, out.data_ptr(), hip.stream_handle()) != 0
    assert lib.cube_subword_pool(ptrs, 33, P * (E + pad), E + pad, E, sp[0].data_ptr(), sp[1].data_ptr(), sp[2].data_ptr(),
                                 n_words, out.data_ptr(), hip.stream_handle()) != 0
