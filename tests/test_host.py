"""CPU tests of the host side: featurisation against the reference's golden collate output,
data-model round trips, weight packing invariants, and that the C-ABI library loads and exports
every symbol include/cube_hip.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from cube_amd import hip, synth
from cube_amd.io_utils.config import ParserConfig
from cube_amd.io_utils.encodings import Encodings
from cube_amd.io_utils.objects import Document
from cube_amd.networks import packing
from cube_amd.networks.utils import MorphoCollate, unpack
from tests import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_collate_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "g1_collate.npz"))
    _, enc, _ = cases.parser_model(cases.TINY)
    doc = cases.document(cases.TINY, enc, "g1")
    out = MorphoCollate(enc).collate_fn(doc.sentences)
    for k in ("x_sent", "x_lang_sent", "x_word", "x_word_case", "x_lang_word", "x_sent_len", "x_word_len",
              "x_sent_masks", "x_word_masks"):
        assert out[k].numpy().dtype == g[k].dtype, k
        np.testing.assert_array_equal(out[k].numpy(), g[k], err_msg=k)
    for i, e in enumerate(out["x_word_embeddings"]):
        np.testing.assert_array_equal(e.numpy(), g["emb%d" % i])
    assert set(out) >= {"y_upos", "y_xpos", "y_attrs"}


def test_unpack_matches_reference_semantics():
    data = torch.arange(12, dtype=torch.float32).view(6, 2)
    out = unpack(data, [2, 0, 3, 1], 4)
    assert out.shape == (4, 4, 2)
    np.testing.assert_array_equal(out[0, :2].numpy(), data[:2].numpy())
    assert (out[1] == 0).all() and (out[0, 2:] == 0).all()
    np.testing.assert_array_equal(out[2, :3].numpy(), data[2:5].numpy())


def test_encodings_and_config_round_trip(tmp_path):
    enc = synth.make_encodings(n_chars=40, n_words=50, n_upos=6, n_xpos=7, n_attrs=8, n_labels=9, num_langs=3)
    f = str(tmp_path / "x.encodings")
    enc.save(f)
    e2 = Encodings()
    e2.load(f)
    assert e2.num_langs == 3 and e2.char2int == enc.char2int and e2.word2int == enc.word2int
    assert e2.labels == enc.labels and e2.upos_list == enc.upos_list and e2.attrs_list == enc.attrs_list
    cfg = ParserConfig()
    cfg.rnn_size = 123
    cfg.lm_model = "dummy:x"
    c = str(tmp_path / "p.config")
    cfg.save(c)
    c2 = ParserConfig(filename=c)
    assert c2.rnn_size == 123 and c2.lm_model == "dummy:x" and c2.char_layers == 5


def test_document_conllu_round_trip(tmp_path, golden_dir):
    src = os.path.join(golden_dir, "g6_process_tiny_bs4.conllu")
    text = open(src, encoding="utf-8").read()
    f = tmp_path / "d.conllu"
    f.write_text(text + "\n\n", encoding="utf-8")
    doc = Document(filename=str(f), lang_id=2)
    assert str(doc) == text
    assert all(s.lang_id == 2 for s in doc.sentences) and len(doc.sentences) == 32
    assert isinstance(doc.sentences[0].words[0].head, int)


def test_gate_packing_is_a_permutation():
    for half in (16, 64, 256):
        order = packing.gate_column_order(half)
        valid = order[order >= 0]
        assert sorted(valid.tolist()) == list(range(2 * half))
        # tanh and sigmoid halves of one channel sit 32 columns apart inside a 64-column wave tile
        pos = {int(c): n for n, c in enumerate(order) if c >= 0}
        for ch in range(half):
            assert pos[half + ch] - pos[ch] == 32 and pos[ch] % 64 < 32


def test_pack_parser_shapes():
    cfg, enc, sd = cases.parser_model(cases.TINY)
    sizes = dict(upos=len(enc.upos2int), xpos=len(enc.xpos2int), attrs=len(enc.attrs2int), labels=len(enc.label2int))
    pk, lay = packing.pack_parser(sd, cfg, sizes, list(cases.TINY["ext"]))
    for cv in pk["char_convs"] + pk["word_convs"]:
        assert cv["W"].hi.shape == (cv["N"], cv["taps"] * cv["Kc"]) and cv["N"] % 128 == 0 and cv["Kc"] % 32 == 0
        assert cv["W"].hi.dtype == torch.bfloat16 and cv["W"].lo.shape == cv["W"].hi.shape
    rows = packing.lstm_gate_rows(cfg.rnn_size)
    assert sorted(rows[rows >= 0].tolist()) == list(range(4 * cfg.rnn_size))
    assert pk["lstm"][0]["whh"].hi.numel() == 2 * len(rows) * packing.p32(cfg.rnn_size)
    # hi + lo reproduces the fp32 weight to ~2^-17 relative
    w = sd["_pre_out.linear_layer.weight"]
    back = pk["pre_out"]["W"].float()[:w.shape[0], :w.shape[1]]
    assert float((back - w).abs().max() / w.abs().max()) < 2e-5
    # LSTM bias = b_ih + b_hh, in packed gate order
    b = (sd["_rnn.bias_ih_l0"] + sd["_rnn.bias_hh_l0"]).numpy()
    got = pk["lstm"][0]["ih"]["bias"].numpy()[:len(rows)]
    np.testing.assert_allclose(got[rows >= 0], b[rows[rows >= 0]], rtol=0, atol=0)


def test_library_exports_every_declared_symbol():
    """Every function declared in include/cube_hip.h is exported by the built library and bound
    by cube_amd.hip (build first: python -c 'import __graft_entry__ as g; g.build()')."""
    header = open(os.path.join(ROOT, "include", "cube_hip.h")).read()
    declared = set(re.findall(r"\b(cube_[a-z0-9_]+)\s*\(", header))
    declared = {d for d in declared if d not in ("cube_gemm_args",)}
    assert declared == set(hip.SIGNATURES), declared ^ set(hip.SIGNATURES)
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.cube_abi_version.restype = ctypes.c_int
    assert lib.cube_abi_version() == hip.ABI_VERSION


def test_parser_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from cube_amd.networks.parser import Parser
    cfg, enc, sd = cases.parser_model(cases.TINY)
    p = Parser(cfg, enc, ext_word_emb=list(cases.TINY["ext"]))
    p.load_state_dict(sd)
    p.to("cpu")
    doc = cases.document(cases.TINY, enc, "single")
    with pytest.raises(hip.CubeHipError):
        p.process(doc, MorphoCollate(enc), batch_size=4)


def test_char_layout_covers_what_the_masked_mean_needs():
    """Parser._char_layout: every word keeps, in layout k, at least min(W, len + 2*(n-k)) positions
    (so n Conv1d(k=5) layers later positions < len are exact), groups tile the rows without gaps,
    word_rows point at each word's first row, and > 31 distinct lengths still fit 32 groups."""
    from cube_amd.networks.parser import Parser
    from tests import cases
    cfg, enc, sd = cases.parser_model(cases.TINY)
    p = Parser(cfg, enc, ext_word_emb=list(cases.TINY["ext"]))
    n = cfg.char_layers
    rng = np.random.default_rng(3)
    for W, lens in ((14, rng.integers(1, 15, size=200)), (40, rng.integers(1, 41, size=500)), (1, np.ones(5, dtype=np.int64)),
                    (9, np.array([9, 0, 3]))):
        lens = np.asarray(lens, dtype=np.int64)
        ch = p._char_layout(lens, W)
        order = ch["order"]
        assert sorted(order.tolist()) == list(range(len(lens)))
        assert len(ch["groups"]) == n and ch["groups"][0].n <= 32
        for l, cg in enumerate(ch["groups"]):
            rin = rout = wd = 0
            for g in range(cg.n):
                assert (cg.row0_in[g], cg.row0_out[g], cg.word0[g]) == (rin, rout, wd)
                nw = cg.n_rows[g] // cg.seq_in[g]
                lg = lens[order[wd:wd + nw]]
                assert cg.seq_in[g] >= min(W, int(lg.max()) + 2 * (n - l)) and cg.seq_in[g] <= W
                assert cg.seq_out[g] >= min(W, max(int(lg.max()), 1) + 2 * (n - l - 1))
                assert cg.seq_out[g] == (ch["groups"][l + 1].seq_in[g] if l + 1 < n else cg.seq_out[g])
                np.testing.assert_array_equal(ch["word_rows"][l, wd:wd + nw], rout + np.arange(nw) * cg.seq_out[g])
                rin, rout, wd = rin + cg.n_rows[g], rout + nw * cg.seq_out[g], wd + nw
            assert (rin, rout, wd) == (ch["rows"][l], ch["rows"][l + 1], len(lens))


def test_abi_rejects_bad_arguments():
    """Every entry point validates its arguments before touching the device and reports through
    the return code + cube_last_error() (no GPU needed: nothing is launched)."""
    lib = hip.load()
    P = 0x10000  # a non-null, 16-byte aligned stand-in pointer: never dereferenced on these paths

    def err(code):
        assert code < 0
        msg = lib.cube_last_error().decode()
        assert msg.startswith("cube_"), msg
        return msg

    def gemm(**kw):
        a = hip.GemmArgs()
        a.A, a.W, a.lda, a.M, a.seq_len, a.taps, a.Kc, a.N = P, P, 64, 128, 128, 1, 64, 128
        a.epilogue, a.out0, a.ldo0, a.n_valid = hip.EPI_F32, P, 128, 128
        for k, v in kw.items():
            setattr(a, k, v)
        return lib.cube_gemm_bf16(ctypes.byref(a), None)

    assert "Kc" in err(gemm(Kc=33))
    assert "N=" in err(gemm(N=100))
    assert "taps" in err(gemm(taps=4))
    assert "A_lo and W_lo" in err(gemm(A_lo=P))
    assert "lda" in err(gemm(lda=60))
    assert "aligned" in err(gemm(A=P + 2))
    assert "seq_len" in err(gemm(taps=5, seq_len=100))
    assert "out0" in err(gemm(ldo0=64))
    assert "unknown epilogue" in err(gemm(epilogue=9))
    assert "bias" in err(gemm(epilogue=hip.EPI_GATE, n_valid=64))
    assert "rowmap" in err(gemm(epilogue=hip.EPI_GATE, n_valid=64, bias=P, rowmap=P))
    assert "bias_tab" in err(gemm(bias_tab=P))
    cg = hip.ConvGroups()
    cg.n = 40
    assert "sequence groups" in err(gemm(epilogue=hip.EPI_GATE, n_valid=64, bias=P, groups=ctypes.pointer(cg)))
    cg.n = 1
    cg.n_rows[0], cg.seq_in[0], cg.seq_out[0] = 100, 7, 5  # 100 rows are not whole sequences of 7
    assert "sequence group" in err(gemm(epilogue=hip.EPI_GATE, n_valid=64, bias=P, taps=5, groups=ctypes.pointer(cg)))
    assert gemm(M=0) == 0  # nothing to do is not an error

    assert "columns" in err(lib.cube_gather_rows(P, None, 64, 60, 8, P, None, 8, None, None, 4, None))
    layers = (ctypes.c_void_p * 9)(*([P] * 9))
    assert "n_layers" in err(lib.cube_masked_mean(layers, None, None, 9, 64, P, 4, 3, 64, P, None, 64, None, None))
    assert "bad args" in err(lib.cube_argmax_rows(P, 16, 10, 8, 4, P, None))
    assert "rnn_size" in err(lib.cube_bilstm_recurrent(P, P, None, 4, 5, 300, P, None, 640, 0, None))
    assert "whh_lo" in err(lib.cube_bilstm_recurrent(P, P, P, 4, 5, 200, P, None, 448, 0, None))
    assert "D/columns" in err(lib.cube_biaffine_arc(P, None, 128, P, None, 256, 128, 100, 2, 9, P, P, None))
    assert "longer than" in err(lib.cube_mst_decode(P, P, 1, 2048, P, P, P, 1 << 40, None))
    assert "workspace too small" in err(lib.cube_mst_decode(P, P, 4, 16, P, P, P, 16, None))
    assert lib.cube_mst_workspace_bytes(4, 200) > lib.cube_mst_workspace_bytes(4, 100) > 0
    assert "D/columns" in err(lib.cube_label_score(P, None, 448, 224, 288, P, None, P, 10, P, P, 2, 3, P, P, None))
    assert "bad args" in err(lib.cube_ext_finish(P, 8, 2, 3, 2, 2, 1, 1, 0, P, None, 64, 0, None, None, 0, 0, None))  # root = 2
    assert "parts" in err(lib.cube_ext_finish(P, 8, 2, 3, 1, 2, 3, 3, 4, P, None, 64, 0, None, None, 0, 0, None))  # 3 parts, rows of 8


def test_host_helper_matches_numpy_collate(monkeypatch):
    """`_cube_host` (csrc/host_collate.c) only replaces two bulk copies: the collate result with it is
    identical to the NumPy-only path (which the golden G1 vectors pin), and it rejects rows it cannot
    stack (the collate then falls back to NumPy's general conversion)."""
    from cube_amd import host
    hm = host.load()
    assert hm is not None, "build() compiles _cube_host.so"
    enc = synth.make_encodings()
    doc = synth.make_document(enc, [3, 1, 17, 8, 1, 30], seed=4, ext_word_emb=(24, 7))
    col = MorphoCollate(enc)
    a = col.collate_fn(doc.sentences)
    monkeypatch.setattr(host, "_mod", None)
    b = col.collate_fn(doc.sentences)
    monkeypatch.undo()
    assert a.keys() == b.keys()
    for k in a:
        if k == "x_word_embeddings":
            assert len(a[k]) == len(b[k]) == 2 and all(torch.equal(x, y) and x.dtype == y.dtype for x, y in zip(a[k], b[k]))
        else:
            assert torch.equal(a[k], b[k]) and a[k].dtype == b[k].dtype, k
    rows = [np.arange(5, dtype=np.float32) + i for i in range(4)]
    out = np.empty((4, 5), dtype=np.float32)
    hm.stack_rows(rows, out)
    assert np.array_equal(out, np.stack(rows))
    for bad in (rows[:3] + [np.arange(6, dtype=np.float32)],            # another length
                rows[:3] + [np.arange(5, dtype=np.int32)],              # another element type, same bytes
                rows[:3] + [np.arange(10, dtype=np.float32)[::2]],      # not contiguous
                rows[:3] + [[0.0] * 5]):                                # not a buffer
        with pytest.raises((ValueError, TypeError, BufferError)):
            hm.stack_rows(bad, np.empty((4, 5), dtype=np.float32))
    o = np.zeros((3, 4), dtype=np.int64)
    hm.fill_ragged(np.arange(1, 6, dtype=np.int64), np.array([2, 0, 3], dtype=np.int64), o, 4)
    assert o.tolist() == [[1, 2, 0, 0], [0, 0, 0, 0], [3, 4, 5, 0]]
    for flat, lens in ((np.arange(5), [2, 0, 2]), (np.arange(5), [5, 0, 0]), (np.arange(5), [-1, 3, 3])):
        with pytest.raises(ValueError):
            hm.fill_ragged(np.asarray(flat, dtype=np.int64), np.asarray(lens, dtype=np.int64), np.zeros((3, 4), dtype=np.int64), 4)
    # words whose embeddings are plain lists (not arrays) still collate
    for s in doc.sentences:
        for w in s.words:
            w.emb = [e.tolist() for e in w.emb]
    c = col.collate_fn(doc.sentences)
    assert all(torch.allclose(x.float(), y.float()) for x, y in zip(c["x_word_embeddings"], a["x_word_embeddings"]))


def test_pack_gated_conv_int8_planes():
    """int8x2 weight planes of cube_gemm_i8x2 (include/cube_hip.h): slice ranges, 64-channel chunk layout and
    the per-column scale reproduce the BN-folded fp32 weights to 15 bits; layers whose input width is not a
    multiple of 64 channels get no int8 planes (the host then stays in bf16x3)."""
    g = torch.Generator().manual_seed(5)
    half, cin, lang = 64, 128, 16
    sd = {"c.0.conv.weight": torch.randn((2 * half, cin + lang, 5), generator=g) * 0.05,
          "c.0.conv.bias": torch.randn(2 * half, generator=g) * 0.1,
          "c.1.weight": torch.rand(2 * half, generator=g) + 0.5, "c.1.bias": torch.randn(2 * half, generator=g) * 0.1,
          "c.1.running_mean": torch.randn(2 * half, generator=g) * 0.1, "c.1.running_var": torch.rand(2 * half, generator=g) + 0.5}
    table = torch.randn((3, lang), generator=g)
    pk = packing.pack_gated_conv(sd, "c", np.arange(cin), double_first=half, bcast=(cin + np.arange(lang), table), int8=True)
    hi, lo = pk["W8"].hi.to(torch.int32), pk["W8"].lo.to(torch.int32)
    assert pk["W8"].hi.dtype == torch.int8 and hi.shape == (pk["N"], 5 * pk["Kc"])
    assert int(hi.abs().max()) <= 127 and int(lo.min()) >= -128 and int(lo.max()) <= 127
    # dequantised, un-chunked (64-channel chunks, taps of a chunk adjacent) == the bf16x3 weights un-chunked (32)
    N, Kc = pk["N"], pk["Kc"]
    w8 = (hi * (1 << hip.I8_SHIFT) + lo).double() * (pk["w_scale"].double() * hip.I8_ONE / float(1 << hip.I8_SHIFT))[:, None]
    w8 = w8.view(N, Kc // 64, 5, 64).permute(0, 2, 1, 3).reshape(N, 5, Kc)
    wf = pk["W"].float().double().view(N, Kc // 32, 5, 32).permute(0, 2, 1, 3).reshape(N, 5, Kc)
    rowmax = wf.abs().amax(dim=(1, 2)).clamp(min=1e-30)
    assert float(((w8 - wf).abs().amax(dim=(1, 2)) / rowmax).max()) < 2.0 ** -15
    # the whole-tensor scale: one unit of (256*HH + X) with activations in units of 1/I8_ONE
    np.testing.assert_allclose(pk["w_scale"].numpy(), (rowmax * float(1 << hip.I8_SHIFT) / hip.I8_ONE ** 2).float().numpy(), rtol=1e-3)
    assert "W8" not in packing.pack_gated_conv(sd, "c", np.arange(32), int8=True)  # Kc = 32


def test_fixed_point_weight_guards():
    """The packer admits a layer to the fixed-point kernels only if its quantised weights stay within a relative
    dot-product error (packing.I8_CONV_WEIGHT_ERR / LSTM8_WEIGHT_ERR): every layer of the random-init full-size model
    passes with room; rows with outliers 60x beyond the bulk are refused (the layer then runs in bf16x3)."""
    cfg, enc, sd = cases.parser_model(cases.FULL)
    from cube_amd.networks.parser import Parser
    p = Parser(cfg, enc, ext_word_emb=[300])
    p.load_state_dict(sd)
    pk = p._pk_cpu
    assert all("W8" in cv for cv in pk["char_convs"][1:]) and all("W8" in cv for cv in pk["word_convs"][1:])
    assert all(ly["whh8"] is not None for ly in pk["lstm"])
    g = torch.Generator().manual_seed(9)
    half, cin = 64, 128
    w = torch.randn((2 * half, cin, 5), generator=g) * 0.05
    sdc = {"c.0.conv.weight": w, "c.0.conv.bias": torch.zeros(2 * half),
           "c.1.weight": torch.ones(2 * half), "c.1.bias": torch.zeros(2 * half),
           "c.1.running_mean": torch.zeros(2 * half), "c.1.running_var": torch.ones(2 * half)}
    ok = packing.pack_gated_conv(sdc, "c", np.arange(cin), int8=True)
    assert "W8" in ok
    w8 = (ok["W8"].hi.to(torch.int32) * (1 << hip.I8_SHIFT) + ok["W8"].lo.to(torch.int32)).double() * (ok["w_scale"].double() * hip.I8_ONE / float(1 << hip.I8_SHIFT))[:, None]
    wf = ok["W"].float().double().view(ok["N"], cin // 32, 5, 32).permute(0, 2, 1, 3).reshape(ok["N"], -1)
    w8 = w8.view(ok["N"], cin // 64, 5, 64).permute(0, 2, 1, 3).reshape(ok["N"], -1)
    assert 1e-5 < packing.quant_row_error(wf, w8) < 6e-5   # Gaussian rows: ~4 sigma / (32512 * sqrt(12))
    w[7, 3, 2] = 3.0                                         # one weight 60 sigma out
    bad = packing.pack_gated_conv(sdc, "c", np.arange(cin), int8=True)
    assert "W8" not in bad and bad["int8_refused"] > packing.I8_CONV_WEIGHT_ERR
    H = 200
    G = pk["lstm"][0]["G"]
    hh = [(torch.rand((G, H), generator=g) * 2 - 1) / np.sqrt(H) for _ in range(2)]
    assert packing.pack_lstm_i8(hh, H) is not None
    hh[1][5, 17] = 4.0
    assert packing.pack_lstm_i8(hh, H) is None


def _host_dicts(p, col, sents):
    """(C featuriser, collate_fn + _prepare_host) for one batch."""
    p.dedup_words = False  # one row per word, as collate_fn has them (the dedup lists: test_c_featurizer_dedup_lists)
    fast = p._featurize(list(sents), col)
    assert fast is not None, "the C featuriser refused a batch in the fast format"
    X = col.collate_fn(list(sents), targets=False)
    return fast, p._prepare_host(X)


def _assert_same_host_batch(fast, slow):
    for k in ("B", "L", "T", "N", "W"):
        assert fast[k] == slow[k], k
    assert list(fast["sizes"]) == list(slow["sizes"])
    assert list(fast["char_rows"]) == list(slow["char_rows"])
    np.testing.assert_array_equal(fast["packed"], slow["packed"])
    assert fast["packed"].dtype == np.int32
    assert len(fast["embs"]) == len(slow["embs"])
    for a, b in zip(fast["embs"], slow["embs"]):
        np.testing.assert_array_equal(a, b)
    assert len(fast["conv_groups"]) == len(slow["conv_groups"])
    for ga, gb in zip(fast["conv_groups"], slow["conv_groups"]):
        assert ga.n == gb.n
        for f in ("row0_in", "n_rows", "seq_in", "row0_out", "seq_out", "word0"):
            assert list(getattr(ga, f))[:ga.n] == list(getattr(gb, f))[:gb.n], f


@pytest.mark.parametrize("which", ["parser", "tagger"])
def test_c_featurizer_equals_collate_plus_prepare(which):
    """csrc/host_collate.c `featurize` (one pass over the Sentence objects) gives bit-identical arrays to
    MorphoCollate.collate_fn (pinned to the reference by G1, utils.py:232-330) followed by the host half of
    `_prepare`: packed indices, ragged character-CNN groups, padded embedding rows.  Ragged sentences, words
    longer than the 31 length groups, upper / symbol / non-ASCII / astral characters, unknown words, K = 2
    embedding layers, and twice in a row (the staging buffers are reused)."""
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.tagger import Tagger
    if which == "parser":
        case = cases.TINY
        cfg, enc, sd = cases.parser_model(case)
        p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    else:
        case = cases.TINY_TAGGER
        cfg, enc, sd = cases.tagger_model(case)
        p = Tagger(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    col = MorphoCollate(enc)
    doc = cases.document(case, enc, "g6")
    w = doc.sentences[3].words
    w[0].word = "Ünïcode"
    w[-1].word = "x" * 45            # beyond the last length group
    doc.sentences[5].words[0].word = "\U0001F600ok中"   # astral + CJK: unknown characters
    doc.sentences[7].words[0].word = "ǅ"   # title-case letter: lower != upper != itself
    for sents in (doc.sentences, doc.sentences[:4], doc.sentences[3:4], doc.sentences[10:27]):
        fast, slow = _host_dicts(p, col, sents)
        _assert_same_host_batch(fast, slow)
        p._staging_release(fast["staging"])
    # not the fast format (float64 vectors): the caller must fall back
    doc.sentences[0].words[0].emb = [e.astype(np.float64) for e in doc.sentences[0].words[0].emb]
    assert p._featurize(list(doc.sentences[:2]), col) is None
    h = p._host_batch(list(doc.sentences[:2]), col)
    assert h["B"] == 2
    # a str subclass may run Python code in .lower() / __hash__: not the fast format either, same arrays through Python
    class Shout(str):
        def lower(self):
            return str.lower(self)
    doc2 = cases.document(case, enc, "g6")
    ref = p._host_batch(list(doc2.sentences[4:7]), col)
    ref = dict(packed=ref["packed"].copy(), st=p._staging_release(ref["staging"]))
    doc2.sentences[5].words[1].word = Shout(doc2.sentences[5].words[1].word)
    assert p._featurize(list(doc2.sentences[4:7]), col) is None
    assert not any(st["busy"] for st in p._staging)
    h2 = p._host_batch(list(doc2.sentences[4:7]), col)
    np.testing.assert_array_equal(np.sort(h2["packed"]), np.sort(ref["packed"]))


def test_featurizer_errors_give_their_staging_set_back():
    """A word that is not a str makes `featurize` raise ValueError: more such errors than the ring has sets
    (`_N_STAGING`) must leave the parser usable — the pinned set goes back on every exception, and `_drain`
    frees what an exception between featurisation and upload left marked busy."""
    from cube_amd.networks.parser import Parser
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    col = MorphoCollate(enc)
    doc = cases.document(case, enc, "g6")
    good = list(doc.sentences[:4])
    bad = list(cases.document(case, enc, "g6").sentences[4:8])
    bad[1].words[0].word = 12345
    for _ in range(p._N_STAGING + 3):
        with pytest.raises((ValueError, TypeError, AttributeError)):
            p._host_batch(bad, col)
    assert not any(st["busy"] for st in p._staging)
    fast, slow = _host_dicts(p, col, good)
    _assert_same_host_batch(fast, slow)
    # sets abandoned between featurisation and upload (an exception in the caller): `_drain` frees them
    held = [p._featurize(good, col) for _ in range(p._N_STAGING - 1)]
    assert all(h is not None for h in held) and all(st["busy"] for st in p._staging)
    p._drain() if hasattr(p, "_drain") else None
    assert not any(st["busy"] for st in p._staging)
    assert p._featurize(good, col) is not None


def test_c_token_table_equals_word2int_lookup():
    """The C-side raw token -> word id table of `featurize` (host_collate.c, WidTable) against utils.py:284-289
    (word2int[word.lower()], <UNK> otherwise): 40 000 distinct tokens so that the table grows several times, tokens
    longer than the 12 inline code points (arena), mixed case (the id belongs to the RAW token), non-ASCII, and a second
    pass that answers everything from the table."""
    from cube_amd import host
    from cube_amd.networks.parser import Parser
    hm = host.load()
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    col = MorphoCollate(enc)
    rng = np.random.default_rng(17)
    vocab = [w for w in enc.word2int if len(w) > 1][:200]
    doc = synth.make_document(enc, [40] * 1000, seed=5, ext_word_emb=tuple(case["ext"]))
    alphabet = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZäöüßÉ中"
    for s_ in doc.sentences:
        for w in s_.words:
            r = rng.random()
            if r < 0.1:
                v = vocab[int(rng.integers(len(vocab)))]
                w.word = v.upper() if rng.random() < 0.5 else v
            else:
                w.word = "".join(alphabet[int(k)] for k in rng.integers(0, len(alphabet), size=int(rng.integers(1, 30))))
    unk = enc.word2int["<UNK>"]
    for rep in range(2):
        for a in range(0, 1000, 250):
            sents = doc.sentences[a:a + 250]
            fast = p._featurize(list(sents), col)
            assert fast is not None
            off = int(np.sum([(x + 3) // 4 * 4 for x in fast["sizes"][:5]]))
            T = fast["T"]
            got = fast["packed"][off:off + len(sents) * T].reshape(len(sents), T)[:, 1:]
            exp = np.array([[enc.word2int.get(w.word.lower(), unk) for w in s_.words] for s_ in sents])
            np.testing.assert_array_equal(got, exp)
            p._staging_release(fast["staging"])
    entries, slots = hm.wid_table_size(col._wid_table)
    assert entries > 30000 and slots >= 2 * entries and slots > (1 << 16)
    assert (exp != unk).any() and (exp == unk).any()


def test_c_featurizer_thread_pool_equals_serial_sweeps():
    """featurize's object pass on the thread pool (batches of >= 1024 words in the plain format) against its serial
    sweeps (`_host_threads = 1`) and against collate_fn + `_prepare_host`: bit-identical arrays -- on a plain batch, with
    code points the character table has not seen (they come back in the same order), with a 3000-character word, with new
    tokens in every call, and with one Word of another class in the batch (the pool declines, the serial sweeps run)."""
    from cube_amd.networks.parser import Parser
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    col_a, col_b = MorphoCollate(enc), MorphoCollate(enc)

    def parser(threads):
        q = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
        q.load_state_dict(sd)
        q._host_threads = threads
        q.dedup_words = True
        return q
    pa, pb = parser(8), parser(1)
    doc = synth.make_document(enc, list(np.random.default_rng(3).integers(1, 40, size=300)), seed=9, ext_word_emb=tuple(case["ext"]))
    assert sum(len(s_.words) for s_ in doc.sentences) >= 4096
    doc.sentences[7].words[0].word = "\u00c5ngstr\u00f6m\U0001F600"   # code points outside the table
    doc.sentences[150].words[-1].word = "Qx" * 1500
    doc.sentences[299].words[0].word = "\u4e2d\u6587"

    def same(sents):
        fa, fb = pa._featurize(list(sents), col_a), pb._featurize(list(sents), col_b)
        assert fa is not None and fb is not None
        _assert_same_host_batch(fa, fb)
        pa._staging_release(fa["staging"])
        pb._staging_release(fb["staging"])
        return fa
    same(doc.sentences)
    same(doc.sentences[100:260])
    pa.dedup_words = pb.dedup_words = False
    fa = same(doc.sentences)
    slow = pb._prepare_host(col_b.collate_fn(list(doc.sentences), targets=False))
    fa = pa._featurize(list(doc.sentences), col_a)
    _assert_same_host_batch(fa, slow)
    pa._staging_release(fa["staging"])

    class Other:   # no __slots__: the getattr path, serial
        pass
    o = Other()
    o.word, o.emb = doc.sentences[5].words[2].word, doc.sentences[5].words[2].emb
    keep = doc.sentences[5].words[2]
    doc.sentences[5].words[2] = o
    same(doc.sentences)
    doc.sentences[5].words[2] = keep


def test_c_featurizer_very_long_words():
    """Words of thousands of characters: the featuriser's code-point buffer (sized for 8 characters per word) grows
    while the strings are being read, and the caller's index buffer is asked for again at its real size."""
    from cube_amd.networks.parser import Parser
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    col = MorphoCollate(enc)
    doc = cases.document(case, enc, "g6")
    sents = doc.sentences[:3]
    sents[0].words[0].word = "ab" * 3000
    sents[2].words[-1].word = "Zz" * 2500
    fast, slow = _host_dicts(p, col, sents)
    assert fast["W"] == 6000
    _assert_same_host_batch(fast, slow)
    p._staging_release(fast["staging"])
    fast, slow = _host_dicts(p, col, doc.sentences[3:6])
    _assert_same_host_batch(fast, slow)
    p._staging_release(fast["staging"])


def test_c_featurizer_full_config_and_generic_objects():
    """Default ParserConfig (first character layer as table look-ups: combined char x case ids) and plain
    objects without __slots__ (what the reference's own Word / Sentence classes are)."""
    from cube_amd.networks.parser import Parser
    cfg, enc, sd = cases.parser_model(cases.FULL)
    p = Parser(cfg, enc, ext_word_emb=[300])
    p.load_state_dict(sd)
    col = MorphoCollate(enc)
    doc = synth.make_document(enc, [32] * 24 + [5, 1, 17], seed=77, ext_word_emb=(300,))
    fast, slow = _host_dicts(p, col, doc.sentences)
    _assert_same_host_batch(fast, slow)
    p._staging_release(fast["staging"])

    class W:   # no __slots__
        pass

    class S:
        pass
    plain = []
    for s in doc.sentences[:5]:
        t = S()
        t.lang_id, t.words = s.lang_id, []
        for w in s.words:
            v = W()
            v.word, v.emb = w.word, w.emb
            t.words.append(v)
        plain.append(t)
    p.dedup_words = False
    fast2 = p._featurize(plain, col)
    slow2 = p._prepare_host(col.collate_fn(doc.sentences[:5], targets=False))
    _assert_same_host_batch(fast2, slow2)
    p._staging_release(fast2["staging"])


def test_c_featurizer_dedup_lists():
    """`dedup_words`: the per-word arrays hold each (spelling, language) pair once, in the order the one-row-per-word
    arrays have their first occurrences; every other occurrence is listed with the XW0 row of the first one."""
    from cube_amd.networks.parser import Parser
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    col = MorphoCollate(enc)
    doc = cases.document(case, enc, "g6")
    sents = list(doc.sentences[:12])
    sents[1].words[0].word = sents[0].words[0].word            # the same word in two sentences
    sents[2].words[-1].word = sents[2].words[0].word           # twice in one sentence
    sents[3].words[0].word = sents[0].words[0].word.upper() + "Q"
    sents[4].words[0].word = sents[3].words[0].word
    p.dedup_words = False
    full = p._featurize(sents, col)
    fsz, fpk = list(full["sizes"]), full["packed"].copy()
    p._staging_release(full["staging"])
    p.dedup_words = True
    dd = p._featurize(sents, col)
    sz, pk = list(dd["sizes"]), dd["packed"]
    assert len(sz) == 12 and len(fsz) == 10 and sz[10] == sz[11] > 0

    def parts(sizes, packed):
        offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sizes])])
        return {n: packed[o:o + s] for n, o, s in zip(p._PART_NAMES, offs[:-1], sizes)}
    a, b = parts(fsz, fpk), parts(sz, pk)
    T = dd["T"]
    words = {}   # XW0 row -> (spelling, language)
    for bi, s_ in enumerate(sents):
        for j, w in enumerate(s_.words):
            words[bi * T + j + p._root] = (w.word, s_.lang_id)
    n_tok, n_u, n_dup = fsz[2], sz[2], sz[10]
    assert dd["N"] == n_u and n_u + n_dup == n_tok == len(words)
    assert len({words[r] for r in b["rowmap_word"]}) == n_u == len(set(words.values()))
    for d, s_ in zip(b["dup_dst"], b["dup_src"]):
        assert words[d] == words[s_] and s_ < d and s_ in set(b["rowmap_word"]) and d not in set(b["rowmap_word"])
    assert sorted(list(b["rowmap_word"]) + list(b["dup_dst"])) == sorted(a["rowmap_word"])
    # per token arrays are untouched; the kept words keep their relative order and their lengths / languages
    for k in ("word_idx", "lang_rows", "root_rows", "lens"):
        np.testing.assert_array_equal(a[k], b[k])
    keep = np.isin(a["rowmap_word"], b["rowmap_word"])
    np.testing.assert_array_equal(a["rowmap_word"][keep], b["rowmap_word"])
    np.testing.assert_array_equal(a["word_len"][keep], b["word_len"])
    np.testing.assert_array_equal(a["lang_word"][keep], b["lang_word"])
    assert list(dd["char_rows"]) < list(full["char_rows"])
    p._staging_release(dd["staging"])


def test_c_write_back_equals_python_loop():
    from cube_amd import host
    from cube_amd.networks.parser import Parser
    hm = host.load()
    assert hm is not None
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    a, b = cases.document(case, enc, "g6"), cases.document(case, enc, "g6")
    B, L = len(a.sentences), max(len(s.words) for s in a.sentences)
    rng = np.random.default_rng(0)
    res = np.stack([rng.integers(0, L + 1, (B, L)), rng.integers(0, len(enc.labels), (B, L)),
                    rng.integers(0, len(enc.upos_list), (B, L)), rng.integers(0, len(enc.xpos_list), (B, L)),
                    rng.integers(0, len(enc.attrs_list), (B, L))]).astype(np.int32)
    status = np.zeros(B, dtype=np.int32)
    p._write_back(a.sentences, 0, res, status)                      # the C call
    p._write_back(tuple(b.sentences), 0, res, status)                # not a list: the Python loop
    assert str(a) == str(b)
    assert isinstance(a.sentences[0].words[0].head, int)
    bad = res.copy()
    bad[2, 0, 0] = len(enc.upos_list)
    with pytest.raises(IndexError):
        hm.write_back(a.sentences, bad, (enc.labels, enc.upos_list, enc.xpos_list, enc.attrs_list))
    with pytest.raises(AssertionError):
        p._write_back(a.sentences, 0, res, np.array([0, 1] + [0] * (B - 2), dtype=np.int32))


def test_batch_geometry_keys_c_equals_python(monkeypatch):
    """sharding.sentence_batch_keys through the C pass (`geometry`) and through the Python loops."""
    from cube_amd import host, sharding
    _, enc, _ = cases.parser_model(cases.TINY)
    rng = np.random.default_rng(4)
    doc = synth.make_document(enc, rng.integers(1, 9, size=203), seed=5, ext_word_emb=(4,), max_word_len=6)

    def cap(k):
        return min(1024, 256 // (k[0] + 1))
    for bs in (4, 7, 32):
        starts = list(range(0, len(doc.sentences), bs))
        fast = sharding.sentence_batch_keys(doc.sentences, starts, bs, cap)
        monkeypatch.setattr(host, "load", lambda: None)
        slow = sharding.sentence_batch_keys(doc.sentences, starts, bs, cap)
        monkeypatch.undo()
        assert fast == slow, bs


def test_geometry_on_the_thread_pool_equals_the_serial_pass():
    """`geometry(sentences, n_threads)` reads plain-format documents on the pool (>= 512 sentences); ragged sentences, an
    empty one, a word of another class (-> the serial pass) and a non-str word (-> the serial pass's TypeError)."""
    from cube_amd import host
    hm = host.load()
    _, enc, _ = cases.parser_model(cases.TINY)
    rng = np.random.default_rng(9)
    doc = synth.make_document(enc, rng.integers(1, 40, size=1500), seed=6, ext_word_emb=(4,), max_word_len=14)
    doc.sentences[17].words = []
    want_len = np.array([len(s.words) for s in doc.sentences], dtype=np.int32)
    want_max = np.array([max((len(w.word) for w in s.words), default=0) for s in doc.sentences], dtype=np.int32)
    for nt in (1, 2, 8):
        slen, wmax = hm.geometry(doc.sentences, nt)
        assert (slen == want_len).all() and (wmax == want_max).all(), nt
    slen, wmax = hm.geometry(doc.sentences)
    assert (slen == want_len).all() and (wmax == want_max).all()

    class Other:
        def __init__(self, word):
            self.word = word
    w = doc.sentences[900].words[0]
    doc.sentences[900].words[0] = Other("x" * 31)
    want_max[900] = max(want_max[900], 31)
    slen, wmax = hm.geometry(doc.sentences, 8)
    assert (slen == want_len).all() and (wmax == want_max).all()
    doc.sentences[900].words[0] = w
    doc.sentences[1200].words[-1].word = 5
    with pytest.raises(TypeError):
        hm.geometry(doc.sentences, 8)


def test_lemma_key_pass_in_c_equals_the_dict_of_tuples():
    """`lemma_keys` / `assign_lemmas` (Lemmatizer.process_bucketed's passes over every word) against the Python pass they
    replace: distinct (word, upos, lang_id) in order of first appearance, words over 25 characters skipped and left
    alone, equal strings that are different objects share a key, an empty sentence, non-str upos values."""
    from cube_amd import host
    hm = host.load()
    _, enc, _ = cases.parser_model(cases.TINY)
    rng = np.random.default_rng(11)
    doc = synth.make_document(enc, rng.integers(1, 30, size=300), seed=8, ext_word_emb=(4,), max_word_len=9)
    sents = doc.sentences
    sents[5].words = []
    tags = ["NOUN", "VERB", None, 3]
    for si, s in enumerate(sents):
        s.lang_id = si % 3
        for wi, w in enumerate(s.words):
            w.upos = tags[(si + wi) % 4]
            w.lemma = "old"
            if (si + wi) % 7 == 0:
                w.word = "".join(["re", "peat"])        # equal text, a new object every time
                w.upos = "".join(["NO", "UN"])
    sents[9].words[0].word = "x" * 26
    sents[10].words[-1].word = "y" * 25

    keys, order, widx = {}, [], []
    for s in sents:
        for w in s.words:
            if len(w.word) <= 25:
                k = (w.word, w.upos, s.lang_id)
                i = keys.setdefault(k, len(order))
                if i == len(order):
                    order.append(k)
                widx.append(i)
    k_word, k_upos, k_lang, lens, idx = hm.lemma_keys(sents, 25)
    assert list(zip(k_word, k_upos, k_lang.tolist())) == order
    assert lens.tolist() == [len(k[0]) for k in order] and idx.tolist() == widx and idx.dtype == np.int32
    lemmas = ["L%d" % i for i in range(len(order))]
    hm.assign_lemmas(sents, idx, lemmas, 25)
    it = iter(widx)
    for s in sents:
        for w in s.words:
            assert w.lemma == ("old" if len(w.word) > 25 else lemmas[next(it)])
    with pytest.raises(ValueError):
        hm.assign_lemmas(sents, idx[:-1].copy(), lemmas, 25)
    with pytest.raises(ValueError):
        hm.assign_lemmas(sents, idx, lemmas[:-1], 25)
    sents[3].words[0].word = 7
    with pytest.raises(TypeError):
        hm.lemma_keys(sents, 25)
    with pytest.raises(TypeError):
        hm.assign_lemmas(sents, idx, lemmas, 25)
