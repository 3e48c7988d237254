"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI of
include/cube_hip.h, against (a) the reference's own outputs in tests/golden/ where the work is
integer/decision work (MST heads), and (b) the oracle in its bf16-contract mode for the
floating-point stages (north_star tolerance: 1e-3 relative on arc/label scores with
fp32-accumulated bf16; decisions equal wherever the margin exceeds the tolerance)."""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3  # north_star: arc/label scores within 1e-3 relative (of the score scale)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from cube_amd import hip
    hip.load()
    return torch.device("cuda:0")


def _bf16_round(x):
    return x.to(torch.bfloat16).to(torch.float32)


# ------------------------------------------------------------------------------------------
# cube_gemm_bf16
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,seq,taps,Kc,N,epi", [
    (300, 300, 1, 64, 128, "f32"), (257, 257, 1, 352, 256, "tanh"), (4 * 33, 33, 5, 96, 128, "relu"),
    (7 * 13, 13, 5, 32, 128, "bf16"), (1000, 1000, 1, 512, 384, "f32"),
])
def test_gemm_plain_epilogues(dev, M, seq, taps, Kc, N, epi):
    from cube_amd import hip
    g = torch.Generator().manual_seed(M + Kc + N)
    lda = Kc + 32
    A = _bf16_round(torch.randn((M, lda), generator=g))
    W = _bf16_round(torch.randn((N, taps * Kc), generator=g) / np.sqrt(taps * Kc))
    bias = torch.randn(N, generator=g)
    # reference: zero-padded taps inside each sequence of `seq` rows
    Ar = A[:, :Kc].view(M // seq, seq, Kc)
    ref = torch.zeros((M // seq, seq, N))
    for t in range(taps):
        d = t - taps // 2
        sh = torch.zeros_like(Ar)
        if d < 0:
            sh[:, -d:] = Ar[:, :seq + d]
        elif d > 0:
            sh[:, :seq - d] = Ar[:, d:]
        else:
            sh = Ar
        ref += sh @ W[:, t * Kc:(t + 1) * Kc].t()
    ref = ref.view(M, N) + bias
    n_valid = N - 5
    Ad, Wd, bd = A.to(torch.bfloat16).to(dev), W.to(torch.bfloat16).to(dev), bias.to(dev)
    code = dict(f32=hip.EPI_F32, bf16=hip.EPI_BF16, tanh=hip.EPI_BF16_TANH, relu=hip.EPI_BF16_RELU)[epi]
    ldo = N + 8
    out = torch.full((M, ldo), 7.0, dtype=torch.float32 if epi == "f32" else torch.bfloat16, device=dev)
    hip.gemm(Ad, lda, M, Kc, Wd, N, bd, code, out0=out, ldo0=ldo, col0=8, taps=taps, seq_len=seq, n_valid=n_valid)
    torch.cuda.synchronize()
    got = out.float().cpu()
    if epi == "tanh":
        ref = torch.tanh(ref)
    if epi == "relu":
        ref = torch.relu(ref)
    tol = 2e-5 if epi == "f32" else 1e-2
    np.testing.assert_allclose(got[:, 8:8 + n_valid].numpy(), ref[:, :n_valid].numpy(), rtol=tol, atol=tol)
    assert (got[:, :8] == 7.0).all() and (got[:, 8 + n_valid:] == 7.0).all()  # nothing outside the valid columns


def test_gemm_gate_epilogue_and_rowmap(dev):
    from cube_amd import hip
    from cube_amd.networks import packing
    g = torch.Generator().manual_seed(5)
    M, seq, Kc, half = 6 * 11, 11, 64, 80
    A = _bf16_round(torch.randn((M, Kc), generator=g))
    w = torch.randn((2 * half, Kc, 5), generator=g) / np.sqrt(5 * Kc)
    sd = {"c.0.conv.weight": w, "c.0.conv.bias": torch.randn(2 * half, generator=g) * 0.1,
          "c.1.weight": torch.rand(2 * half, generator=g) + 0.5, "c.1.bias": torch.randn(2 * half, generator=g) * 0.1,
          "c.1.running_mean": torch.randn(2 * half, generator=g) * 0.1, "c.1.running_var": torch.rand(2 * half, generator=g) + 0.5}
    pk = packing.pack_gated_conv(sd, "c", np.arange(Kc))
    s = sd["c.1.weight"] / torch.sqrt(sd["c.1.running_var"] + 1e-5)
    wf = _bf16_round(w * s[:, None, None])
    bf = (sd["c.0.conv.bias"] - sd["c.1.running_mean"]) * s + sd["c.1.bias"]
    co = torch.nn.functional.conv1d(A.view(M // seq, seq, Kc).permute(0, 2, 1), wf, bf, padding=2)
    gate = (torch.tanh(co[:, :half]) * torch.sigmoid(co[:, half:])).permute(0, 2, 1).reshape(M, half)
    res0 = torch.randn((M, half), generator=g)
    for mode in (0, 1, 2):
        res = res0.clone().to(dev)
        out0 = torch.zeros((M, half + 8), dtype=torch.bfloat16, device=dev)
        out1 = torch.zeros((M, half), dtype=torch.bfloat16, device=dev)
        hip.gemm(A.to(torch.bfloat16).to(dev), Kc, M, pk["Kc"], pk["W"].to(dev), pk["N"], pk["bias"].to(dev),
                 hip.EPI_GATE, out0=out0, ldo0=half + 8, col0=8, taps=5, seq_len=seq, n_valid=half, res=res,
                 ld_res=half, res_mode=mode, out1=out1, ldo1=half, col1=0)
        torch.cuda.synchronize()
        exp_res = gate if mode == 0 else (res0 + gate if mode == 1 else res0 + 2 * gate)
        np.testing.assert_allclose(res.cpu().numpy(), exp_res.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out0[:, 8:].float().cpu().numpy(), gate.numpy(), rtol=1e-2, atol=1e-2)
        np.testing.assert_allclose(out1.float().cpu().numpy(), (exp_res + gate).numpy(), rtol=1e-2, atol=1e-2)
    # rowmap scatter (unpack fused into the epilogue)
    Wl = _bf16_round(torch.randn((128, Kc), generator=g) / 8)
    rowmap = torch.arange(M, dtype=torch.int32) * 2
    rowmap[3] = -1
    out = torch.zeros((2 * M, 128), dtype=torch.float32, device=dev)
    hip.gemm(A.to(torch.bfloat16).to(dev), Kc, M, Kc, Wl.to(torch.bfloat16).to(dev), 128, None, hip.EPI_F32, out0=out,
             ldo0=128, n_valid=128, rowmap=rowmap.to(dev))
    torch.cuda.synchronize()
    ref = A @ Wl.t()
    ref[3] = 0
    np.testing.assert_allclose(out.cpu().numpy()[0::2], ref.numpy(), rtol=1e-4, atol=1e-4)
    assert (out.cpu().numpy()[1::2] == 0).all()


# ------------------------------------------------------------------------------------------
# cube_mst_decode against the REFERENCE's heads (tests/golden/g4_mst.npz)
# ------------------------------------------------------------------------------------------
def test_mst_matches_reference_heads(dev, golden_dir):
    from cube_amd.networks.utils import ChuLiuEdmondsDecoder
    g = np.load(os.path.join(golden_dir, "g4_mst.npz"))
    dec = ChuLiuEdmondsDecoder(dev)
    for key, (probs, lens) in cases.mst_cases().items():
        exp = g[key]
        if exp.ndim == 1:
            with pytest.raises(AssertionError):
                dec.decode(probs, lens)
            continue
        heads = dec.decode(probs, lens)
        for b, h in enumerate(heads):
            np.testing.assert_array_equal(h, exp[b, :len(h)], err_msg="%s[%d]" % (key, b))


# ------------------------------------------------------------------------------------------
# whole path against the bf16-contract oracle
# ------------------------------------------------------------------------------------------
def _run_case(dev, case, doc_key):
    from oracle import parser_oracle
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    cfg, enc, sd = cases.parser_model(case)
    doc = cases.document(case, enc, doc_key)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    p.to(dev)
    X = MorphoCollate(enc).collate_fn(doc.sentences)
    X.pop("y_upos")
    r = p.decode_batch(X)
    torch.cuda.synchronize()
    ora = parser_oracle.OracleParser(cfg, enc, sd, list(case["ext"]), quant="bf16")
    torch.set_num_threads(16)
    with torch.no_grad():
        e = ora.decode_batch(parser_oracle.collate(doc.sentences, enc))
    return p, r, e, ora


def _margin_ok(logits_exp, got_idx, exp_idx, tol):
    """decisions must agree wherever the oracle's top-2 margin exceeds `tol` (absolute)."""
    flat = logits_exp.reshape(-1, logits_exp.shape[-1])
    gi, ei = got_idx.reshape(-1), exp_idx.reshape(-1)
    top2 = np.sort(flat, axis=-1)[:, -2:]
    margin = top2[:, 1] - top2[:, 0]
    bad = (gi != ei) & (margin > tol)
    return int(bad.sum()), float((gi == ei).mean())


@pytest.mark.parametrize("case,doc_key", [
    (cases.TINY, "equal"), (cases.TINY, "ragged"), (cases.TINY, "single"),
    (cases.FULL, "full_8x32"), (cases.FULL, "full_2x128"), (cases.FULL, "ragged"),
])
def test_parser_batch_matches_oracle(dev, case, doc_key):
    p, r, e, _ = _run_case(dev, case, doc_key)
    sl = r["bt"]["lens"].cpu().numpy()
    arc = r["arc_logits"].cpu().numpy()
    assert _rel(arc, e["arc_logits"]) < REL_TOL
    probs = r["arc_probs"].cpu().numpy()
    np.testing.assert_allclose(probs, e["arc_probs"], rtol=5e-3, atol=1e-6)
    for name, key in (("upos_logits", "upos_logits"),):
        assert _rel(r[name].cpu().numpy(), e[key]) < REL_TOL, name
    n_a = p._sizes["attrs"]
    ax = r["ax_logits"].cpu().numpy()
    upos_same = (r["upos"].cpu().numpy() == e["upos"])
    # xpos/attrs are conditioned on the predicted UPOS: compare where it agrees
    assert _rel(ax[:, :, :n_a][upos_same], e["attrs_logits"][upos_same]) < REL_TOL
    assert _rel(ax[:, :, n_a:][upos_same], e["xpos_logits"][upos_same]) < REL_TOL
    mask = np.arange(arc.shape[1])[None, :] < sl[:, None]
    tol_abs = REL_TOL * np.abs(e["upos_logits"]).max()
    nbad, agree = _margin_ok(e["upos_logits"][mask], r["upos"].cpu().numpy()[mask], e["upos"][mask], tol_abs)
    assert nbad == 0 and agree >= 0.999, (nbad, agree)
    # heads: the MST of the device probabilities must equal the oracle's tree
    heads = r["heads"].cpu().numpy()
    exp_heads = np.zeros_like(heads)
    for b, h in enumerate(e["heads"]):
        exp_heads[b, :len(h)] = h
    assert (r["status"].cpu().numpy() == 0).all()
    head_agree = (heads[mask] == exp_heads[mask]).mean()
    assert head_agree >= 0.999, head_agree
    # labels: scores compared where the head agrees
    same = mask & (heads == exp_heads)
    lab = r["label_logits"].cpu().numpy()
    assert _rel(lab[same], e["label_logits"][same]) < REL_TOL
    tol_abs = REL_TOL * np.abs(e["label_logits"]).max()
    nbad, agree = _margin_ok(e["label_logits"][same], r["labels"].cpu().numpy()[same], e["labels"][same], tol_abs)
    assert nbad == 0 and agree >= 0.999, (nbad, agree)


def test_forward_api_outputs(dev):
    """Parser.forward keeps the reference's 8-tuple (parser.py:216) incl. the aux heads."""
    from oracle import parser_oracle
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    doc = cases.document(case, enc, "ragged")
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    p.to(dev)
    X = MorphoCollate(enc).collate_fn(doc.sentences)
    X.pop("y_upos")
    outs = p.forward(X)
    ora = parser_oracle.OracleParser(cfg, enc, sd, list(case["ext"]), quant="bf16")
    with torch.no_grad():
        exp = ora.forward(parser_oracle.collate(doc.sentences, enc))
    names = ("arc", "pre_parsing", "upos", "xpos", "attrs", "aupos", "axpos", "aattrs")
    for n, a, b in zip(names, outs, exp):
        assert tuple(a.shape) == tuple(b.shape), n
        assert _rel(a.cpu().numpy(), b.numpy()) < (REL_TOL if n != "pre_parsing" else 1e-2), n
    heads = [np.array([0] * int(l)) for l in X["x_sent_len"]]
    lab = p._get_labels(outs[1], heads)
    exp_lab = ora.get_labels(exp[1], heads)
    assert _rel(lab.cpu().numpy(), exp_lab.numpy()) < 2e-3


@pytest.mark.parametrize("bs", [4, 32])
def test_process_document_text(dev, bs):
    """Parser.process end to end: CoNLL-U text equal to the oracle's (bf16 contract) and, for the
    decisions that are not within rounding of a tie, to the reference's own text (golden G6)."""
    from oracle import parser_oracle
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    case = cases.TINY
    cfg, enc, sd = cases.parser_model(case)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    p.load_state_dict(sd)
    p.to(dev)
    doc = cases.document(case, enc, "g6")
    p.process(doc, MorphoCollate(enc), batch_size=bs, num_workers=0)
    ora = parser_oracle.OracleParser(cfg, enc, sd, list(case["ext"]), quant="bf16")
    doc2 = cases.document(case, enc, "g6")
    ora.process(doc2, batch_size=bs)
    a, b = str(doc).split("\n"), str(doc2).split("\n")
    assert len(a) == len(b)
    diff = sum(x != y for x, y in zip(a, b))
    assert diff <= max(1, len(a) // 1000), diff
