"""int8x2 fixed-point path of the character CNN (cube_gemm_i8x2, cube_char_conv0_lookup_i8,
cube_masked_mean_i8; include/cube_hip.h) through the C ABI.

* the GEMM against torch.conv1d in float64 on the SAME quantised operands (what remains is the dropped
  lo*lo term, 2^-16 relative, and the 3.1e-5 output step);
* the whole parser with `char_int8` on against the REFERENCE's own outputs (tests/golden), to the same
  bar as the default mode (scores 1e-3 relative, decisions >= 99.9 %).
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
from tests import cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _planes(x):
    """float tensor in [-1, 1] -> (q int32, hi int8, lo int8), q = round(x * I8_ONE) = 256*hi + lo."""
    from cube_amd.hip import I8_ONE, I8_SHIFT
    q = torch.round(x.double() * I8_ONE).to(torch.int32).clamp(-I8_ONE, I8_ONE)
    hi = (q + (1 << (I8_SHIFT - 1))) >> I8_SHIFT
    lo = q - (hi << I8_SHIFT)
    return q, hi.to(torch.int8), lo.to(torch.int8)


@pytest.mark.parametrize("big", [False, True])
def test_gemm_i8x2_ragged_groups_and_bias_table(dev, big):
    from cube_amd import hip
    from cube_amd.networks import packing
    g = torch.Generator().manual_seed(11)
    Kc, nb, half, n_lang = 128, 16, 256 if big else 48, 4
    spec = [(3, 1, 1), (5, 3, 1), (900 if big else 7, 7, 5), (5000 if big else 9, 14, 14), (4, 9, 8)]
    w = torch.randn((2 * half, Kc + nb, 5), generator=g) / np.sqrt(5 * (Kc + nb))
    sd = {"c.0.conv.weight": w, "c.0.conv.bias": torch.randn(2 * half, generator=g) * 0.1,
          "c.1.weight": torch.rand(2 * half, generator=g) + 0.5, "c.1.bias": torch.randn(2 * half, generator=g) * 0.1,
          "c.1.running_mean": torch.randn(2 * half, generator=g) * 0.1, "c.1.running_var": torch.rand(2 * half, generator=g) + 0.5}
    table = torch.randn((n_lang, nb), generator=g)
    pk = packing.pack_gated_conv(sd, "c", np.arange(Kc), bcast=(Kc + np.arange(nb), table), int8=True)
    # the weights the kernel sees, back in reference channel order: un-chunk, un-permute the columns
    N = pk["N"]
    w8 = (pk["W8"].hi.to(torch.int32) * (1 << hip.I8_SHIFT) + pk["W8"].lo.to(torch.int32)).double()
    w8 = w8.view(N, Kc // 64, 5, 64).permute(0, 2, 1, 3).reshape(N, 5, Kc)
    wq = w8 * (pk["w_scale"].double() * hip.I8_ONE / float(1 << hip.I8_SHIFT))[:, None, None]          # [N, taps, Kc] dequantised
    order = packing.gate_column_order(half)
    wref = torch.zeros((2 * half, Kc, 5), dtype=torch.float64)
    wref[torch.from_numpy(order[order >= 0])] = wq[torch.from_numpy(np.where(order >= 0)[0])].permute(0, 2, 1)
    s = sd["c.1.weight"] / torch.sqrt(sd["c.1.running_var"] + 1e-5)
    wf = (w * s[:, None, None]).double()
    assert float((wref - wf[:, :Kc]).abs().max() / wf.abs().max()) < 5e-5  # 16-bit weights
    bf = (sd["c.0.conv.bias"] - sd["c.1.running_mean"]) * s + sd["c.1.bias"]
    cg = hip.ConvGroups()
    cg.n = len(spec)
    rin = rout = wd = 0
    for k, (nw, si, so) in enumerate(spec):
        cg.row0_in[k], cg.n_rows[k], cg.seq_in[k], cg.row0_out[k], cg.seq_out[k], cg.word0[k] = rin, nw * si, si, rout, so, wd
        rin, rout, wd = rin + nw * si, rout + nw * so, wd + nw
    A = torch.rand((rin, Kc), generator=g) * 2 - 1
    A[0, :4] = torch.tensor([1.0, -1.0, 0.0, 0.99999])
    q, ahi, alo = _planes(A)
    Aq = q.double() / hip.I8_ONE
    lang = torch.randint(0, n_lang, (wd,), generator=g)
    exp = torch.zeros((rout, half), dtype=torch.float64)
    wfull = torch.cat([wref, wf[:, Kc:]], 1)
    for k, (nw, si, so) in enumerate(spec):
        a = Aq[cg.row0_in[k]:cg.row0_in[k] + nw * si].view(nw, si, Kc)
        lg = table[lang[cg.word0[k]:cg.word0[k] + nw]].double()[:, None, :].expand(nw, si, nb)
        co = torch.nn.functional.conv1d(torch.cat([a, lg], -1).permute(0, 2, 1), wfull, bf.double(), padding=2)
        gate = (torch.tanh(co[:, :half]) * torch.sigmoid(co[:, half:])).permute(0, 2, 1)[:, :so, :]
        exp[cg.row0_out[k]:cg.row0_out[k] + nw * so] = gate.reshape(nw * so, half)
    ld = (half + 63) // 64 * 64 + 16
    out = hip.Pair(torch.full((rout + 3, ld), 7, dtype=torch.int8, device=dev),
                   torch.full((rout + 3, ld), 7, dtype=torch.int8, device=dev))
    hip.gemm(hip.Pair(ahi.to(dev), alo.to(dev)), Kc, rin, pk["Kc"], packing.to_device(pk["W8"], dev), pk["N"],
             pk["bias"].to(dev), hip.EPI_GATE, out0=out, ldo0=ld, col0=0, taps=5, seq_len=14, n_valid=half,
             bias_tab=pk["bias_tab"].to(dev), row_group=lang.to(torch.int32).to(dev), groups=cg,
             w_scale=pk["w_scale"].to(dev))
    torch.cuda.synchronize()
    hi, lo = out.hi.cpu().to(torch.int32), out.lo.cpu().to(torch.int32)
    assert int(hi[:rout, :half].abs().max()) <= 127 and int(lo[:rout, :half].min()) >= -128 and int(lo[:rout, :half].max()) <= 127
    got = (hi * (1 << hip.I8_SHIFT) + lo).double() / hip.I8_ONE
    np.testing.assert_allclose(got[:rout, :half].numpy(), exp.numpy(), rtol=0, atol=1.5e-4)
    assert (hi[rout:] == 7).all() and (lo[rout:] == 7).all() and (hi[:, (half + 15) // 16 * 16:] == 7).all()  # untouched


def test_gemm_i8x2_rejects_what_it_does_not_support(dev):
    from cube_amd import hip
    z = torch.zeros((64, 64), dtype=torch.int8, device=dev)
    w = hip.Pair(torch.zeros((128, 64), dtype=torch.int8, device=dev), torch.zeros((128, 64), dtype=torch.int8, device=dev))
    w5 = hip.Pair(torch.zeros((128, 320), dtype=torch.int8, device=dev), torch.zeros((128, 320), dtype=torch.int8, device=dev))
    ws = torch.ones(128, device=dev)
    bias = torch.zeros(128, device=dev)
    out = hip.Pair(torch.zeros((64, 64), dtype=torch.int8, device=dev), torch.zeros((64, 64), dtype=torch.int8, device=dev))
    with pytest.raises(hip.CubeHipError):  # Kc must be a multiple of 64
        hip.gemm(hip.Pair(z, z), 64, 64, 32, w, 128, bias, hip.EPI_GATE, out0=out, ldo0=64, n_valid=64, w_scale=ws)
    with pytest.raises(hip.CubeHipError):  # gate epilogue only
        hip.gemm(hip.Pair(z, z), 64, 64, 64, w, 128, bias, hip.EPI_BF16, out0=out, ldo0=64, n_valid=64, w_scale=ws)
    with pytest.raises(hip.CubeHipError, match="res_mode"):  # residual stream: res = g or res += g only
        hip.gemm(hip.Pair(z, z), 64, 64, 64, w5, 128, bias, hip.EPI_GATE, out0=out, ldo0=64, n_valid=64, w_scale=ws,
                 res=torch.zeros((64, 64), device=dev), ld_res=64, res_mode=2, taps=5, seq_len=64)
    with pytest.raises(hip.CubeHipError, match="four channels"):  # rows of res are read four channels at a time
        hip.gemm(hip.Pair(z, z), 64, 64, 64, w5, 128, bias, hip.EPI_GATE, out0=out, ldo0=64, n_valid=64, w_scale=ws,
                 res=torch.zeros((64, 66), device=dev), ld_res=66, res_mode=1, taps=5, seq_len=64)


def test_word_level_int8x2_layers_match_float64(dev):
    """The word-level stack's two int8x2 pieces through the C ABI: (1) cube_gemm_bf16 with `out0_int8` -- bf16x3
    operands, fp32 residual stream, gate values out as int8x2 planes; (2) cube_gemm_i8x2 with the residual stream
    (res += g) and the x + res side outputs.  Both against torch.conv1d in float64 on the operands the kernels see."""
    from cube_amd import hip
    from cube_amd.networks import packing
    g = torch.Generator().manual_seed(5)
    B, T, Kc, half, n_lang, nb = 7, 33, 256, 256, 3, 16
    M = B * T

    def conv_sd(cin):
        w = torch.randn((2 * half, cin, 5), generator=g) / np.sqrt(5 * cin)
        return {"c.0.conv.weight": w, "c.0.conv.bias": torch.randn(2 * half, generator=g) * 0.1,
                "c.1.weight": torch.rand(2 * half, generator=g) + 0.5, "c.1.bias": torch.randn(2 * half, generator=g) * 0.1,
                "c.1.running_mean": torch.randn(2 * half, generator=g) * 0.1, "c.1.running_var": torch.rand(2 * half, generator=g) + 0.5}

    def folded(sd):
        s = sd["c.1.weight"] / torch.sqrt(sd["c.1.running_var"] + 1e-5)
        return (sd["c.0.conv.weight"] * s[:, None, None]).double(), ((sd["c.0.conv.bias"] - sd["c.1.running_mean"]) * s + sd["c.1.bias"]).double()

    def gate(x, wf, bf):  # x [B, T, C] float64 -> [M, half]
        co = torch.nn.functional.conv1d(x.permute(0, 2, 1), wf, bf, padding=2)
        return (torch.tanh(co[:, :half]) * torch.sigmoid(co[:, half:])).permute(0, 2, 1).reshape(M, half)

    # (1) layer 0: bf16x3 in, int8x2 planes + residual out
    sd0 = conv_sd(Kc)
    pk0 = packing.pack_gated_conv(sd0, "c", np.arange(Kc))
    x0 = torch.randn((M, Kc), generator=g)
    a0 = hip.Pair.split(x0)
    x0q = (a0.hi.double() + a0.lo.double()).view(B, T, Kc)
    w0 = packing.to_device(pk0["W"], dev)
    wf0, bf0 = folded(sd0)
    g0 = gate(x0q, wf0, bf0)
    q0 = hip.Pair(torch.zeros((M, half), dtype=torch.int8, device=dev), torch.zeros((M, half), dtype=torch.int8, device=dev))
    res = torch.full((M, half), 3.0, device=dev)
    hip.gemm(a0.to(dev), Kc, M, pk0["Kc"], w0, pk0["N"], pk0["bias"].to(dev), hip.EPI_GATE, out0=q0, ldo0=half, taps=5,
             seq_len=T, n_valid=half, res=res, ld_res=half, res_mode=0, out0_int8=True)
    torch.cuda.synchronize()
    got0 = (q0.hi.cpu().to(torch.int32) * (1 << hip.I8_SHIFT) + q0.lo.cpu().to(torch.int32)).double() / hip.I8_ONE
    np.testing.assert_allclose(got0.numpy(), g0.numpy(), rtol=0, atol=1e-4)          # 3.1e-5 output step + bf16x3
    np.testing.assert_allclose(res.cpu().double().numpy(), g0.numpy(), rtol=0, atol=2e-5)   # res_mode 0: res = g
    # (2) layer 1: int8x2 in, res += g, x + res into two bf16 hi/lo outputs, int8x2 planes out; language bias table
    sd1 = conv_sd(Kc + nb)
    table = torch.randn((n_lang, nb), generator=g)
    pk1 = packing.pack_gated_conv(sd1, "c", np.arange(Kc), bcast=(Kc + np.arange(nb), table), int8=True)
    lang = torch.randint(0, n_lang, (B,), generator=g)
    lang_rows = lang.repeat_interleave(T).to(torch.int32)
    wf1, bf1 = folded(sd1)
    x1 = torch.cat([got0.view(B, T, Kc), table[lang].double()[:, None, :].expand(B, T, nb)], -1)
    g1 = gate(x1, wf1, bf1)   # with the fp64 weights: the 16-bit weight rows add ~5e-6
    q1 = hip.Pair(torch.zeros((M, half), dtype=torch.int8, device=dev), torch.zeros((M, half), dtype=torch.int8, device=dev))
    o1 = hip.Pair(torch.zeros((M, half + 8), dtype=torch.bfloat16, device=dev), torch.zeros((M, half + 8), dtype=torch.bfloat16, device=dev))
    o2 = hip.Pair(torch.zeros((M, half), dtype=torch.bfloat16, device=dev), torch.zeros((M, half), dtype=torch.bfloat16, device=dev))
    res_before = res.cpu().double()
    for out0 in (q1, None):   # the last layer of the stack has no gate output
        res.copy_(res_before.float())
        hip.gemm(q0, half, M, pk1["Kc"], packing.to_device(pk1["W8"], dev), pk1["N"], pk1["bias"].to(dev), hip.EPI_GATE,
                 out0=out0, ldo0=half, taps=5, seq_len=T, n_valid=half, res=res, ld_res=half, res_mode=1,
                 out1=o1, ldo1=half + 8, col1=4, out2=o2, ldo2=half, col2=0, bias_tab=pk1["bias_tab"].to(dev),
                 row_group=lang_rows.to(dev), w_scale=pk1["w_scale"].to(dev))
        torch.cuda.synchronize()
        rs = res_before + g1
        np.testing.assert_allclose(res.cpu().double().numpy(), rs.numpy(), rtol=0, atol=1e-4)
        xs = (rs + g1).numpy()
        np.testing.assert_allclose((o2.hi.cpu().double() + o2.lo.cpu().double()).numpy(), xs, rtol=0, atol=2e-4)
        np.testing.assert_allclose((o1.hi.cpu().double() + o1.lo.cpu().double())[:, 4:4 + half].numpy(), xs, rtol=0, atol=2e-4)
        assert (o1.hi.cpu()[:, :4] == 0).all() and (o1.hi.cpu()[:, 4 + half:] == 0).all()   # untouched
    got1 = (q1.hi.cpu().to(torch.int32) * (1 << hip.I8_SHIFT) + q1.lo.cpu().to(torch.int32)).double() / hip.I8_ONE
    np.testing.assert_allclose(got1.numpy(), g1.numpy(), rtol=0, atol=1.5e-4)


def test_quant_i8x2_is_round_to_nearest_of_the_scaled_value(dev):
    """cube_quant_i8x2 against NumPy: q = rint(clip((hi + lo) * inv_scale, -1, 1) * I8_ONE) = 256*hi + lo, bit for bit,
    padding columns (inv_scale 0) zero, values past the scale clamped."""
    from cube_amd import hip
    g = torch.Generator().manual_seed(11)
    rows, cols, ld_x, ld_q = 77, 96, 104, 112
    x = torch.randn((rows, ld_x), generator=g) * 1.5
    x[3, 5], x[4, 6] = 9.0, -9.0
    pair = hip.Pair.split(x)
    inv = (1.0 / (torch.rand(cols, generator=g) * 3 + 0.5)).float()
    inv[80:] = 0.0
    q = hip.Pair(torch.full((rows, ld_q), 7, dtype=torch.int8, device=dev), torch.full((rows, ld_q), 7, dtype=torch.int8, device=dev))
    pd, invd = pair.to(dev), inv.to(dev)
    hip.check(hip.load().cube_quant_i8x2(pd.hi.data_ptr(), pd.lo.data_ptr(), ld_x, rows, cols, invd.data_ptr(), q.hi.data_ptr(),
                                         q.lo.data_ptr(), ld_q, hip.stream_handle()), "cube_quant_i8x2")
    torch.cuda.synchronize()
    v = (pair.hi.float() + pair.lo.float())[:, :cols] * inv[None, :]          # fp32, as the kernel
    want = np.rint(np.clip(v.numpy(), -1.0, 1.0).astype(np.float32) * np.float32(hip.I8_ONE) ).astype(np.int64)
    # the kernel rounds v * I8_ONE inside ONE fma (no intermediate rounding of the product): allow the half-way cases
    exact = np.clip(v.double().numpy(), -1, 1) * hip.I8_ONE
    got = q.hi.cpu().numpy().astype(np.int64)[:, :cols] * (1 << hip.I8_SHIFT) + q.lo.cpu().numpy().astype(np.int64)[:, :cols]
    assert np.abs(got - exact).max() <= 0.5 + 1e-6 and (np.abs(got - want) <= 1).all()
    assert (got[:, 80:] == 0).all() and got[3, 5] == hip.I8_ONE and got[4, 6] == -hip.I8_ONE
    hi8 = q.hi.cpu().numpy()[:, :cols].astype(np.int64)
    assert np.abs(hi8).max() <= 127
    assert (q.hi.cpu()[:, cols:] == 7).all() and (q.lo.cpu()[:, cols:] == 7).all()   # nothing past `cols` is written


_CASES = [(cases.FULL, "full_8x32", "g3_forward_full.npz"), (cases.FULL, "full_2x128", "g3_forward_full.npz"),
          (cases.FULL, "ragged", "g3_forward_full.npz")]


@pytest.mark.parametrize("i8", [True, False, "word0"])
@pytest.mark.parametrize("case,doc_key,fixture", _CASES)
def test_parser_char_int8_matches_reference(dev, golden_dir, case, doc_key, fixture, i8):
    """`Parser.char_int8` on (the default) and off (bf16x3 in the character CNN too) against the
    reference's own outputs, to the same bar; "word0": the opt-in int8x2 route of the FIRST word-level gated
    convolution as well (cube_quant_i8x2 + cube_gemm_i8x2 with the block scales folded into its weights)."""
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    from tests.test_gpu_parity import REL_TOL, _margin_ok, _rel
    cfg, enc, sd = cases.parser_model(case)
    p = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
    word0, i8 = i8 == "word0", bool(i8)
    p.char_int8 = i8
    p.word0_int8 = word0
    p.load_state_dict(sd)
    p.to(dev)
    doc = cases.document(case, enc, doc_key)
    X = MorphoCollate(enc).collate_fn(doc.sentences)
    X.pop("y_upos")
    r = p.decode_batch(X)
    torch.cuda.synchronize()
    assert ("xc_q1" in p._buf._b) == i8 and ("xc_g1" in p._buf._b) == (not i8), "wrong character-CNN path ran"
    assert ("xw0q" in p._buf._b) == word0, "wrong route of the first word-level convolution ran"
    g = np.load(os.path.join(golden_dir, fixture))
    sl = r["bt"]["lens"].cpu().numpy()
    arc, ref_arc = r["arc_logits"].cpu().numpy(), g["%s/arc_logits" % doc_key]
    assert _rel(arc, ref_arc) < REL_TOL, _rel(arc, ref_arc)
    assert _rel(r["upos_logits"].cpu().numpy(), g["%s/upos" % doc_key]) < REL_TOL
    mask = np.arange(arc.shape[1])[None, :] < sl[:, None]
    got_upos, ref_upos = r["upos"].cpu().numpy(), np.argmax(g["%s/upos" % doc_key], axis=-1)
    nbad, agree = _margin_ok(g["%s/upos" % doc_key][mask], got_upos[mask], ref_upos[mask],
                             REL_TOL * np.abs(g["%s/upos" % doc_key]).max())
    assert nbad == 0 and agree >= 0.999, (nbad, agree)
    heads, ref_heads = r["heads"].cpu().numpy(), g["%s/heads" % doc_key]
    assert (r["status"].cpu().numpy() == 0).all()
    assert (heads[mask] == ref_heads[mask]).mean() >= 0.999
    same = mask & (heads == ref_heads)
    lab, ref_lab = r["label_logits"].cpu().numpy(), g["%s/label_logits" % doc_key]
    assert _rel(lab[same], ref_lab[same]) < REL_TOL, _rel(lab[same], ref_lab[same])
    nbad, agree = _margin_ok(ref_lab[same], r["labels"].cpu().numpy()[same], np.argmax(ref_lab, -1)[same],
                             REL_TOL * np.abs(ref_lab).max())
    assert nbad == 0 and agree >= 0.999, (nbad, agree)
    print("char_int8=%s %s: arc %.2e upos %.2e label %.2e" % (i8, doc_key, _rel(arc, ref_arc),
          _rel(r["upos_logits"].cpu().numpy(), g["%s/upos" % doc_key]), _rel(lab[same], ref_lab[same])))


# ------------------------------------------------------------------------------------------
# cube_bilstm_recurrent_i8: hidden-state weights resident in VGPRs + LDS, 15-bit h x 12-bit W_hh on the int8 MFMAs
# ------------------------------------------------------------------------------------------
def _lstm_i8_reference(gx, whh_dirs, H, B, T, lo_bits):
    """The kernel's arithmetic in NumPy: exact integer products of the quantised operands, fp64 gate math."""
    from cube_amd.hip import LSTM8_HONE as I8_ONE, I8_SHIFT
    wstep = float(1 << lo_bits)
    LSTM8_WONE = 127.0 * wstep
    hstep = float(1 << I8_SHIFT)
    G = whh_dirs[0].shape[0]
    U = G // 64
    out = np.zeros((B, T, 2 * H))
    cols = np.array([[u * 64 + g * 16 + j for u in range(U) for j in range(16)] for g in range(4)])   # [gate][hidden]
    for d, w in enumerate(whh_dirs):
        w = w.double().numpy()
        wmax = np.maximum(np.abs(w).max(axis=1), 1e-30)
        q = np.round(w / wmax[:, None] * LSTM8_WONE)
        whi = np.floor((q + wstep / 2) / wstep)
        wlo = q - wstep * whi
        scale = wmax / (I8_ONE * LSTM8_WONE)
        h = np.zeros((B, U * 16))
        c = np.zeros((B, U * 16))
        for s in range(T):
            t = T - 1 - s if d else s
            hq = np.clip(np.round(h * I8_ONE), -I8_ONE, I8_ONE)
            hhi = np.floor((hq + hstep / 2) / hstep)
            hlo = hq - hstep * hhi
            K = w.shape[1]
            acc = hstep * wstep * (hhi[:, :K] @ whi.T) + hstep * (hhi[:, :K] @ wlo.T) + wstep * (hlo[:, :K] @ whi.T)
            pre = acc * scale[None, :] + gx[:, t, d * G:(d + 1) * G]
            i, f, g, o = (pre[:, cols[k]] for k in range(4))
            sig = lambda x: 1.0 / (1.0 + np.exp(-x))
            c = sig(f) * c + sig(i) * np.tanh(g)
            h = sig(o) * np.tanh(c)
            out[:, t, d * H:(d + 1) * H] = h[:, :H]
    return out


@pytest.mark.parametrize("H,B,T,lo_bits", [(200, 37, 33, None), (200, 37, 33, 4), (16, 5, 7, None), (64, 16, 12, None),
                                           (64, 16, 12, 4), (96, 3, 129, None), (40, 50, 4, 6), (40, 50, 4, 4), (208, 20, 6, 6),
                                           (256, 9, 5, None), (224, 4, 9, None)])
def test_bilstm_i8_matches_its_fixed_point_contract(dev, H, B, T, lo_bits):
    """The resident-weight recurrence against a NumPy restatement of the same fixed-point arithmetic (exact integer
    products, fp64 gates): every shape class of the kernel — K = 64 steps with and without the K = 32 tail, one wave to
    sixteen, a batch that does not fill the last 16-sentence workgroup, input projections through LDS (H <= 208) and
    through registers, six-bit low slices of W_hh (three bytes for four values; the default up to H = 208) and nibbles.  A
    wrong MFMA operand map or bit order cannot pass: the weights are asymmetric random."""
    from cube_amd import hip
    from cube_amd.networks import packing
    lib = hip.load()
    rng = np.random.default_rng(H * 1000 + B)
    rows1 = packing.lstm_gate_rows(H)
    G = len(rows1)
    v = np.where(rows1 >= 0)[0]
    whh = []
    for d in range(2):
        w = torch.from_numpy(rng.uniform(-1, 1, size=(4 * H, H)) / np.sqrt(H)).float()
        wp = torch.zeros((G, H), dtype=torch.float32)
        wp[torch.from_numpy(v)] = w[torch.from_numpy(rows1[v])]
        whh.append(wp)
    pk = packing.pack_lstm_i8(whh, H, lo_bits)
    assert lib.cube_bilstm_lo_bits(H) == packing.lstm8_lo_bits(H) and pk["lo_bits"] == (lo_bits or packing.lstm8_lo_bits(H))
    gx = np.zeros((B, T, 2 * G), dtype=np.float32)
    gx[:, :, np.concatenate([v, G + v])] = rng.standard_normal((B, T, 2 * len(v))).astype(np.float32)
    ld_out, col0 = 2 * H + 24, 8
    out = torch.zeros((B * T, ld_out), dtype=torch.bfloat16, device=dev)
    out_lo = torch.zeros_like(out)
    d_gx = torch.from_numpy(gx).to(dev)
    d_hi, d_lo, d_sc = pk["hi"].to(dev), pk["lo"].to(dev), pk["scale"].to(dev)
    hip.check(lib.cube_bilstm_recurrent_i8(d_gx.data_ptr(), d_hi.data_ptr(), d_lo.data_ptr(), d_sc.data_ptr(), B, T, H,
                                           pk["lo_bits"], out.data_ptr(), out_lo.data_ptr(), ld_out, col0, hip.stream_handle()),
              "cube_bilstm_recurrent_i8")
    torch.cuda.synchronize()
    got = (out.float() + out_lo.float()).cpu().numpy().reshape(B, T, ld_out)
    assert (got[:, :, :col0] == 0).all() and (got[:, :, col0 + 2 * H:] == 0).all()
    exp = _lstm_i8_reference(gx.astype(np.float64), whh, H, B, T, pk["lo_bits"])
    err = np.abs(got[:, :, col0:col0 + 2 * H] - exp).max()
    assert err < 1.5e-4, err   # one step of the 16-bit h grid is 3e-5; a flipped rounding propagates below that


def test_parser_lstm_int8_and_bf16x3_both_match_reference(dev, golden_dir):
    """rnn_out is not an output of the path, but the arc and label scores read nothing else: with the BiLSTM in int8x2
    (default) and in bf16x3 (`lstm_int8 = False`) they stay within 1e-3 of the REFERENCE's own (G3) and of each other."""
    from tests.test_gpu_parity import _model, _rel
    from cube_amd.networks.utils import MorphoCollate
    g = np.load(os.path.join(golden_dir, "g3_forward_full.npz"))
    p, cfg, enc, sd = _model(dev, cases.FULL, "bf16x3")
    res = {}
    try:
        for flag in (True, False):
            p.lstm_int8 = flag
            for key in ("full_8x32", "full_2x128"):
                doc = cases.document(cases.FULL, enc, key)
                X = MorphoCollate(enc).collate_fn(doc.sentences, targets=False)
                r = p.decode_batch(X)
                torch.cuda.synchronize()
                arc = r["arc_logits"].cpu().numpy().copy()
                assert _rel(arc, g["%s/arc_logits" % key]) < 1e-3, (flag, key, _rel(arc, g["%s/arc_logits" % key]))
                res[(flag, key)] = arc
    finally:
        p.lstm_int8 = True
    for key in ("full_8x32", "full_2x128"):
        assert _rel(res[(True, key)], res[(False, key)]) < 2e-4


def test_outlier_weights_fall_back_to_bf16x3(dev):
    """A checkpoint whose weights have outliers far beyond the bulk (here: one weight 60 sigma out in a character
    layer, a word-level layer and a recurrent matrix) is refused by the packer's fixed-point guards
    (packing.I8_CONV_WEIGHT_ERR / LSTM8_WEIGHT_ERR): the three stages then run the bf16x3 kernels -- bit-identical to
    switching the fixed-point paths off by hand -- and stay within the tolerance of the fp32 oracle."""
    from oracle import parser_oracle
    from tests.test_gpu_parity import _rel
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    case = cases.FULL
    cfg, enc, sd = cases.parser_model(case)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["_word_net._convolutions_char.2.0.conv.weight"][5, 40, 1] = 3.0
    sd["_convs.3.0.conv.weight"][100, 7, 4] = -3.0
    sd["_rnn.weight_hh_l1"][17, 3] = 4.0

    def build(**flags):
        q = Parser(cfg, enc, ext_word_emb=list(case["ext"]))
        q.load_state_dict(sd)
        for k, v in flags.items():
            setattr(q, k, v)
        return q.to(dev)

    guarded, manual = build(), build(char_int8=False, word_int8=False)
    pk = guarded._pk_cpu
    assert "W8" not in pk["char_convs"][2] and "W8" not in pk["word_convs"][3] and pk["lstm"][1]["whh8"] is None
    assert pk["lstm"][0]["whh8"] is not None and "W8" in pk["char_convs"][1]  # the other layers were admitted
    doc = cases.document(case, enc, "full_8x32")
    X = MorphoCollate(enc).collate_fn(doc.sentences, targets=False)
    a = guarded.decode_batch(X)
    arc_a, packed_a = a["arc_logits"].cpu().numpy().copy(), a["packed"].cpu().numpy().copy()
    b = manual.decode_batch(X)
    # (layers 0 and 2 of the BiLSTM keep the fixed-point recurrence in both models)
    assert np.array_equal(arc_a, b["arc_logits"].cpu().numpy()) and np.array_equal(packed_a, b["packed"].cpu().numpy())
    ora = parser_oracle.OracleParser(cfg, enc, sd, list(case["ext"]), quant="fp32")
    with torch.no_grad():
        e = ora.decode_batch(X)
    assert _rel(arc_a, e["arc_logits"]) < 1e-3
