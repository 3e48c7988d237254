"""GPU parity tests of the lemmatizer row, through the C ABI: the HIP path against
the reference's own outputs (tests/golden/g9_*, produced by make_golden.py g9 with the reference
imported) and against the oracle (oracle/lemmatizer_oracle.py) on the same seeded inputs.

Greedy decoding feeds every arg-max back, so a near-tie flips the rest of that word: sequences
must be identical wherever the oracle's top-2 margin at the first differing step exceeds the
tolerance; logits of the first steps within 1e-3 relative (the north-star tolerance)."""
import os

import numpy as np
import pytest
import torch

from tests import cases

pytestmark = pytest.mark.gpu

REL_TOL = 1e-3
CASES = {"small": cases.LEMMA_SMALL, "full": cases.LEMMA}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from cube_amd import hip
    hip.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def g9(golden_dir):
    return np.load(os.path.join(golden_dir, "g9_lemmatizer.npz"))


_models = {}


def _model(tag, dev, precision="bf16x3"):
    key = (tag, precision)
    if key not in _models:
        from cube_amd.networks.lemmatizer import Lemmatizer
        cfg, enc, sd = cases.lemmatizer_model(CASES[tag])
        m = Lemmatizer(cfg, enc)
        m.precision = precision
        m.load_state_dict(sd)
        m.to(dev)
        _models[key] = (m, cfg, enc, sd)
    return _models[key]


def _batch(enc, dk, n):
    from cube_amd.networks.lemmatizer import LemmaDataset, Word2TargetCollate
    ds = LemmaDataset(cases.lemma_document(enc, dk), for_training=False)
    return Word2TargetCollate(enc).collate_fn([ds[i] for i in range(min(n, len(ds)))], targets=False)


ROUNDOFF_RATIO = 2.0 ** 7   # unit round-off of the bf16x3 operands (2^-17) over that of the reference's fp32 (2^-24)


def _compare_sequences(got_logits, got_k, ref_c, ref_k, l32, l64):
    """Greedy arg-max histories (from the HIP path's logits [words, steps, C]) against the reference's `ref_c`.  A word may
    differ only where north_star lets a decision differ:
    (a) the fp32 oracle's top-2 margin at the FIRST differing step t is below 1e-3 of the logit scale (a near-tie), or
    (b) the word's history is decided by round-off: up to step t the HIP path's logits deviate from the fp32 oracle's
        (same history) by no more than 2 x ROUNDOFF_RATIO x what the oracle's OWN arithmetic is uncertain by there -- the
        largest difference between its fp32 and float64 runs (`l32`, `l64`) over the steps <= t.  The fed-back decoder of
        a random-init model is chaotic for the words that never emit the stop symbol: every round-off grows by the same
        factor (measured: x 3000 in 60 steps, for the fp32 oracle against float64 exactly as for the HIP path against
        the fp32 oracle, a constant ratio of ~75 between the two), so this bounds the implementation's precision at its
        16-17 bits without pretending that the 1e-3 score tolerance survives hundreds of fed-back steps.
    Everything else must be identical; no count of tolerated words.  -> (differing words, of them by rule b)."""
    got_c = got_logits.argmax(-1)
    assert got_c.shape == ref_c.shape, (got_c.shape, ref_c.shape)
    same = got_c == ref_c
    bad = np.where(~same.all(1))[0]
    n_b = 0
    for w in bad:
        t = int(np.argmin(same[w]))
        top = np.sort(l32[w, t].astype(np.float64))[-2:]
        scale = max(1.0, float(np.abs(l32[w, t]).max()))
        near_tie = top[1] - top[0] < REL_TOL * scale
        msg = ""
        if not near_tie:
            n = min(t + 1, l64.shape[1])
            sc = np.maximum(np.abs(l32[w, :n]).max(axis=-1), 1.0)
            e_ref = float((np.abs(l64[w, :n] - l32[w, :n]).max(axis=-1) / sc).max())
            e_dev = float((np.abs(got_logits[w, :n] - l32[w, :n]).max(axis=-1) / sc).max())
            msg = "; deviation %.2g = %.0f x the oracle's own fp32 / float64 difference %.2g" % (e_dev, e_dev / e_ref, e_ref)
            assert n == t + 1 and e_dev <= 2 * ROUNDOFF_RATIO * e_ref, (int(w), t, top.tolist(), e_dev, e_ref)
            n_b += 1
        print("word %d differs from step %d: top-2 margin %.3g (logit scale %.3g)%s" % (w, t, top[1] - top[0], scale, msg))
    ok = same.all(1)
    # the case head (4 logits per step, not fed back) has near-ties of its own
    assert (got_k[ok] != ref_k[ok]).mean() <= 1e-3
    return len(bad), n_b


def test_collate_matches_reference_golden(g9):
    """Word2TargetCollate of this package == the reference's tensors (CPU work, kept with its consumer)."""
    for tag in ("small", "full"):
        _, enc, _ = cases.lemmatizer_model(CASES[tag])
        for dk, n in (("ragged", 37), ("g6", 128)):
            X = _batch(enc, dk, n)
            for k in ("x_char", "x_case", "x_lang", "x_upos"):
                assert np.array_equal(X[k].numpy(), g9["%s/%s/%s" % (tag, dk, k)]), (tag, dk, k)


@pytest.mark.parametrize("tag", ["small", "full"])
def test_encoder_matches_oracle(dev, tag):
    """encoder_output (lemmatizer.py:89-124) and the encoder half of the attention conv."""
    from oracle import lemmatizer_oracle as lo
    m, cfg, enc, sd = _model(tag, dev)
    X = _batch(enc, "ragged", 37)
    ed = m._encode(X)
    lay = m._lay
    B, S = ed["B"], ed["S"]
    got = ed["enc"].float()[:B * S * lay.enc_w].view(B, S, lay.enc_w).cpu()
    ref = lo.OracleLemmatizer(cfg, enc, sd).encode(X)
    perm = torch.from_numpy(lay.enc_perm())
    cols = torch.where(perm >= 0)[0]
    g = got[:, :, cols]
    r = ref[:, :, perm[cols]]
    assert float((g - r).abs().max() / r.abs().max()) < REL_TOL
    assert float(got[:, :, torch.where(perm < 0)[0]].abs().max()) == 0.0
    # e_enc = Conv1d(k=5) over the encoder channels + bias
    w = sd["_attention.attn.conv.weight"].float()[:, cfg.decoder_size:, :]
    e_ref = torch.nn.functional.conv1d(ref.permute(0, 2, 1), w, sd["_attention.attn.conv.bias"].float(), padding=2)
    e_got = ed["e_enc"][:B * S * ed["ld_e"]].view(B, S, ed["ld_e"])[:, :, :cfg.att_proj_size].cpu()
    assert float((e_got - e_ref.permute(0, 2, 1)).abs().max() / e_ref.abs().max()) < REL_TOL


@pytest.mark.parametrize("tag,dk,n", [("full", "ragged", 37), ("full", "g6", 128), ("small", "ragged", 37), ("small", "g6", 128)])
def test_forward_matches_reference_golden(dev, g9, tag, dk, n):
    """Lemmatizer.forward (greedy decode, EOL stop / 20*S failsafe) against the reference's outputs."""
    m, cfg, enc, sd = _model(tag, dev)
    X = _batch(enc, dk, n)
    yc, yk = m.forward(X)
    ref_c, ref_k = g9["%s/%s/char" % (tag, dk)], g9["%s/%s/case" % (tag, dk)]
    yc, yk = yc.cpu().numpy(), yk.cpu().numpy()
    assert yc.shape == ref_c.shape + (len(enc.char2int) + 2,) and yk.shape == ref_k.shape + (4,)
    l16 = g9["%s/%s/char_logits16" % (tag, dk)]
    k16 = g9["%s/%s/case_logits16" % (tag, dk)]
    got_c, got_k = yc.argmax(-1), yk.argmax(-1)
    # logits of the first steps, for the words whose character history still agrees with the reference
    nstep = min(16, ref_c.shape[1])
    for t in range(nstep):
        agree = (got_c[:, :t] == ref_c[:, :t]).all(axis=1)
        assert agree.mean() >= 0.99  # measured: no word leaves the reference's history before step 20
        d = np.abs(yc[agree, t] - l16[agree, t]).max() / np.abs(l16[agree, t]).max()
        assert d < REL_TOL, (t, d)
        d = np.abs(yk[agree, t] - k16[agree, t]).max() / max(np.abs(k16[agree, t]).max(), 1e-6)
        assert d < REL_TOL, (t, d)
    # margins at a first differing step: the oracle's logits for the same character history
    from oracle import lemmatizer_oracle as lo
    oc_all = lo.OracleLemmatizer(cfg, enc, sd).forward(X)[0].numpy()
    assert np.array_equal(oc_all.argmax(-1), ref_c) or tag == "small"
    l64 = lo.OracleLemmatizer(cfg, enc, sd, dtype=torch.float64).forward(X)[0].numpy() if (got_c != ref_c).any() else oc_all
    nbad, n_chaotic = _compare_sequences(yc, got_k, ref_c, ref_k, oc_all, l64)
    print("%s/%s: %d of %d words differ from the reference (%d near-ties, %d decided by round-off)"
          % (tag, dk, nbad, len(ref_c), nbad - n_chaotic, n_chaotic))


@pytest.mark.parametrize("tag,bs", [("full", 32), ("full", 4), ("small", 32)])
def test_process_matches_reference_golden(dev, golden_dir, tag, bs):
    """Lemmatizer.process end to end: lemma strings of the g6 document at the reference's default
    batch size (4) and at 32 (results depend on batch composition: S = longest word + 2)."""
    from cube_amd.networks.lemmatizer import Word2TargetCollate
    m, cfg, enc, sd = _model(tag, dev)
    doc = cases.lemma_document(enc, "g6", max_word_len=6)
    m.process(doc, Word2TargetCollate(enc), batch_size=bs, num_workers=0)
    got = [w.word + "\t" + w.lemma for s in doc.sentences for w in s.words]
    with open(os.path.join(golden_dir, "g9_lemmas_%s_bs%d.txt" % (tag, bs)), encoding="utf-8") as f:
        ref = f.read().split("\n")[:-1]
    assert len(got) == len(ref)
    bad = [i for i, (a, b) in enumerate(zip(got, ref)) if a != b]
    print("%s bs%d: %d of %d lemmas differ" % (tag, bs, len(bad), len(ref)))
    if not bad:
        return
    # A lemma that differs is held to the rule of `_compare_sequences` in the reference batch it was decoded in (in-order
    # batches of `bs` words, lemmatizer.py:189-193): near-tie at the first differing step, or a word the float64 oracle
    # shows to be decided by round-off.  No count of tolerated lemmas.
    from oracle import lemmatizer_oracle as lo
    from cube_amd.networks.lemmatizer import LemmaDataset
    ds = LemmaDataset(cases.lemma_document(enc, "g6", max_word_len=6), for_training=False)
    col = Word2TargetCollate(enc)
    o32 = lo.OracleLemmatizer(cfg, enc, sd)
    for a in sorted({i // bs * bs for i in bad}):
        ids = list(range(a, min(a + bs, len(ds))))
        X = col.collate_fn([ds[i] for i in ids], targets=False)
        yc, yk = m.forward(X)
        yc, yk = yc.cpu().numpy(), yk.cpu().numpy()
        oc, ok = yc.argmax(-1), yk.argmax(-1)
        strings = m._strings(oc, ok)
        assert [ds[i]["word"] + "\t" + sl for i, sl in zip(ids, strings)] == [got[i] for i in ids]   # process == this batch decoded
        l32, k32 = (t.numpy() for t in o32.forward(X))
        ref_c = l32.argmax(-1)
        assert oc.shape == ref_c.shape
        # the fp32 oracle IS the reference on this batch: its strings are the stored ones
        ref_strings = m._strings(ref_c, k32.argmax(-1))
        l64 = lo.OracleLemmatizer(cfg, enc, sd, dtype=torch.float64).forward(X)[0].numpy()
        n64 = min(l64.shape[1], ref_c.shape[1])
        keep = []
        for j, (i, r) in enumerate(zip(ids, ref_strings)):
            if ds[i]["word"] + "\t" + r == ref[i]:
                keep.append(j)
            else:   # (the oracle's own BLAS order already leaves the stored text: float64 must show the word undecided too)
                assert (l64[j, :n64].argmax(-1) != ref_c[j, :n64]).any(), (a, j)
        _compare_sequences(yc[keep], ok[keep], ref_c[keep], ok[keep], l32[keep], l64[keep])


def test_bf16_single_pass_mode_runs_and_mostly_agrees(dev, g9):
    """precision='bf16' (one MFMA pass): same pipeline; early characters agree with the reference for
    most words (no stronger claim: greedy feedback amplifies bf16 rounding)."""
    m, cfg, enc, sd = _model("full", dev, "bf16")
    oc, ok = m.decode_batch(_batch(enc, "ragged", 37))
    ref = g9["full/ragged/char"]
    assert oc.shape == ref.shape
    assert (oc[:, :3] == ref[:, :3]).all(axis=1).mean() > 0.8


def test_long_words_and_single_word_batches(dev):
    """S up to 27 (25-character words), a batch of one word, and the IndexError of the reference when a
    word longer than 25 characters is skipped by the dataset (lemmatizer.py:225-229)."""
    from oracle import lemmatizer_oracle as lo
    from cube_amd.networks.lemmatizer import LemmaDataset, Word2TargetCollate
    m, cfg, enc, sd = _model("full", dev)
    doc = cases.lemma_document(enc, "ragged")
    doc.sentences[0].words[0].word = "ab" * 12 + "c"      # 25 characters
    ds = LemmaDataset(doc, for_training=False)
    col = Word2TargetCollate(enc)
    orc = lo.OracleLemmatizer(cfg, enc, sd)
    for items in ([ds[0]], [ds[i] for i in range(6)]):
        X = col.collate_fn(items, targets=False)
        oc, ok = m.decode_batch(X)
        yc, yk = orc.forward(X)
        rc, rk = yc.argmax(-1).numpy(), yk.argmax(-1).numpy()
        assert np.array_equal(oc, rc) and np.array_equal(ok, rk)
    doc.sentences[0].words[0].word = "x" * 26
    with pytest.raises(IndexError):
        m.process(doc, col, batch_size=8)


def test_cube_api_runs_parser_then_lemmatizer(dev, tmp_path):
    """CubeObj with `-trf-lemmatizer.{config,encodings,best}` in the model directory: parser, then
    lemmatizer on the parser's UPOS tags (reference cube/api.py:82-90,124-126) == the two stages by hand."""
    import yaml
    from cube_amd import synth
    from cube_amd.api import CubeObj
    from cube_amd.networks.lemmatizer import Word2TargetCollate
    from cube_amd.networks.lm import LMHelperPrecomputed
    from cube_amd.networks.parser import Parser
    from cube_amd.networks.utils import MorphoCollate
    pcase = cases.TINY
    pcfg, penc, psd = cases.parser_model(pcase)
    lcfg, lenc, lsd = cases.lemmatizer_model(cases.LEMMA_SMALL)   # same TINY vocabulary
    base = str(tmp_path / "en")
    with open(base + "-trf-tokenizer.yaml", "w") as f:
        yaml.safe_dump({"language_codes": synth.EN_LANGUAGE_CODES, "language_map": {"en": synth.EN_LANGUAGE_CODES[3]}}, f)
    for name, cfg, enc, sd in (("parser", pcfg, penc, psd), ("lemmatizer", lcfg, lenc, lsd)):
        cfg.save("%s-trf-%s.config" % (base, name))
        enc.save("%s-trf-%s.encodings" % (base, name))
        torch.save({"state_dict": sd}, "%s-trf-%s.best" % (base, name))
    cube = CubeObj(base, device=str(dev), lang="en", lm_helper=LMHelperPrecomputed(list(pcase["ext"])))
    a = cases.document(pcase, penc, "ragged")
    b = cases.document(pcase, penc, "ragged")
    cube(a)
    p = Parser(pcfg, penc, ext_word_emb=list(pcase["ext"]))
    p.load_state_dict(psd)
    p.to(dev)
    p.process(b, MorphoCollate(penc), num_workers=0)
    m, _, _, _ = _model("small", dev)
    m.process(b, Word2TargetCollate(lenc), num_workers=0)
    assert str(a) == str(b)
    assert any(w.lemma not in ("_", "") for s in a.sentences for w in s.words)


def test_process_bucketed_is_batch_independent(dev):
    """process_bucketed: equal-length batches of distinct (word, UPOS, language) keys.  Each word's
    lemma equals what the same word gives alone in a batch of one (rows of the kernels do not
    interact), duplicates share one result, over-long words keep their lemma."""
    from cube_amd.networks.lemmatizer import Word2TargetCollate
    m, cfg, enc, sd = _model("full", dev)
    col = Word2TargetCollate(enc)
    doc = cases.lemma_document(enc, "ragged", max_word_len=5)
    words = [w for s in doc.sentences for w in s.words]
    words[3].word, words[3].upos = words[0].word, words[0].upos          # a duplicate key ...
    doc.sentences[0].lang_id = doc.sentences[1].lang_id = 2             # ... (words 0 and 3 are in sentence 0)
    words[5].word, words[5].lemma = "y" * 30, "keep-me"
    m.process_bucketed(doc, col, max_words=7)
    assert words[5].lemma == "keep-me"
    assert words[0].lemma == words[3].lemma
    for sent in doc.sentences[:3]:
        for w in sent.words:
            if len(w.word) > 25:
                continue
            X = col.collate_fn([{"word": w.word, "upos": w.upos, "lang_id": sent.lang_id, "target": ""}], targets=False)
            assert m._strings(*m.decode_batch(X))[0] == w.lemma


def test_merged_reference_batches_do_not_change_lemmas(dev):
    """Lemmatizer.process with merge=True == merge=False at the reference's default batch size of 4 words."""
    from cube_amd.networks.lemmatizer import Word2TargetCollate
    m, cfg, enc, sd = _model("full", dev)
    col = Word2TargetCollate(enc)
    a = cases.lemma_document(enc, "g6", max_word_len=4)
    b = cases.lemma_document(enc, "g6", max_word_len=4)
    m.max_steps = 24   # bound the test: a random decoder runs into the 20*S failsafe otherwise
    try:
        m.process(a, col, batch_size=4, merge=True)
        m.process(b, col, batch_size=4, merge=False)
    finally:
        m.max_steps = None
    assert [w.lemma for s in a.sentences for w in s.words] == [w.lemma for s in b.sentences for w in s.words]


@pytest.mark.parametrize("units,M,K", [(300, 37, 384), (300, 1300, 640), (16, 5, 32), (200, 300, 96)])
def test_lstm_epilogue_equals_gemm_then_cell(dev, units, M, K):
    """CUBE_EPI_LSTM (the cell update of lemmatizer.py:131,149 in the epilogue of the gate GEMM, gate rows packed per
    include/cube_hip.h) gives the same bits as CUBE_EPI_F32 followed by cube_lstm_cell: h, the bf16 copy of c, and c."""
    from cube_amd import hip
    from cube_amd.networks import packing
    lib = hip.load()
    g = torch.Generator().manual_seed(units * 1000 + M)
    W = torch.randn((4 * units, K), generator=g) * 0.2
    b = torch.randn(4 * units, generator=g) * 0.3
    x = torch.randn((M, K), generator=g)
    c_prev = torch.randn((M, units), generator=g)
    ng = (4 * units + 255) // 256 * 256
    Wd, bd = torch.zeros((ng, K)), torch.zeros(ng)
    Wd[:4 * units], bd[:4 * units] = W, b
    perm = packing.lstm_epilogue_rows(units)
    ngf = (len(perm) + 255) // 256 * 256
    Wf, bf = torch.zeros((ngf, K)), torch.zeros(ngf)
    Wf[:len(perm)][torch.from_numpy(perm >= 0)] = W[torch.from_numpy(perm[perm >= 0])]
    bf[:len(perm)][torch.from_numpy(perm >= 0)] = b[torch.from_numpy(perm[perm >= 0])]
    up = lambda t: hip.Pair.split(t).to(dev)
    A, Wd_p, Wf_p = up(x), up(Wd), up(Wf)
    bd, bf = bd.to(dev), bf.to(dev)
    ld_h = packing.p32(units) * 2
    res = {}
    for fused in (False, True):
        c = c_prev.clone().to(dev).contiguous()
        h = hip.Pair(torch.zeros((M, ld_h), dtype=torch.bfloat16, device=dev), torch.zeros((M, ld_h), dtype=torch.bfloat16, device=dev))
        col_h, col_c = 0, packing.p32(units)
        if fused:
            hip.gemm(A, K, M, K, Wf_p, ngf, bf, hip.EPI_LSTM, out0=h, ldo0=ld_h, col0=col_h, n_valid=units, res=c, ld_res=units,
                     out1=h, ldo1=ld_h, col1=col_c)
        else:
            gates = torch.zeros((M, ng), dtype=torch.float32, device=dev)
            hip.gemm(A, K, M, K, Wd_p, ng, bd, hip.EPI_F32, out0=gates, ldo0=ng, n_valid=ng)
            hip.check(lib.cube_lstm_cell(gates.data_ptr(), ng, units, c.data_ptr(), M, h.hi.data_ptr(), h.lo.data_ptr(), ld_h, col_h,
                                         h.hi.data_ptr(), h.lo.data_ptr(), col_c, hip.stream_handle()), "cube_lstm_cell")
        torch.cuda.synchronize()
        res[fused] = (c.cpu(), h.hi.cpu(), h.lo.cpu())
    bits = lambda t: t.view(torch.int16) if t.dtype == torch.bfloat16 else t.view(torch.int32)
    for a_, b_ in zip(res[False], res[True]):
        assert torch.equal(bits(a_), bits(b_))
    # and the update itself against torch's formula (fp32 grade: the operands are hi/lo bf16 pairs)
    gt = x.double() @ W.double().T + b.double()
    i_, f_, g_, o_ = [gt[:, k * units:(k + 1) * units] for k in range(4)]
    c_ref = torch.sigmoid(f_) * c_prev.double() + torch.sigmoid(i_) * torch.tanh(g_)
    h_ref = torch.sigmoid(o_) * torch.tanh(c_ref)
    h_got = (res[True][1].float() + res[True][2].float())[:, :units].double()
    assert float((res[True][0].double() - c_ref).abs().max()) < 1e-3 * float(c_ref.abs().max())
    assert float((h_got - h_ref).abs().max()) < 1e-3
    # an output that overlaps the columns of A is refused
    with pytest.raises(hip.CubeHipError):
        hip.gemm(A, K, M, K, Wf_p, ngf, bf, hip.EPI_LSTM, out0=A, ldo0=K, col0=0, n_valid=min(units, K), res=c, ld_res=units)


@pytest.mark.parametrize("tag", ["small", "full"])
def test_fused_cell_decoder_equals_separate_cell_kernel(dev, tag):
    """Lemmatizer.forward with the cell updates in the GEMM epilogues (two decoder-state buffers taken in turn) == the
    same with CUBE_EPI_F32 + cube_lstm_cell on one buffer: every logit of every step, bit for bit."""
    m, cfg, enc, sd = _model(tag, dev)
    X = _batch(enc, "ragged", 37)
    try:
        m.fused_cell = True
        yc1, yk1 = m.forward(X)
        m.fused_cell = False
        yc0, yk0 = m.forward(X)
    finally:
        m.fused_cell = True
    assert torch.equal(yc0, yc1) and torch.equal(yk0, yk1)
