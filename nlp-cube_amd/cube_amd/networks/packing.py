"""Load-time weight packing for the HIP parser/tagger path.

Turns a reference `state_dict` (key names of cube/networks/parser.py:23-95 and
tagger.py:22-67) into the layouts the kernels read:

* eval-mode BatchNorm1d folded into the conv weight/bias (modules.py:438-446), conv weights
  re-laid as [N][tap*Kc + c] bf16 (n-major, K contiguous, Kc = channels padded to 32) with the
  output channels permuted so that the tanh half and the sigmoid half of one channel land in
  the same lane of the MFMA accumulator (`cube_gemm_bf16`, CUBE_EPI_GATE);
* WordGram's `x = x + skip` doubling (modules.py:467-468) folded into the weights of char
  layers >= 1 (exact: a power of two);
* Linear weights zero-padded to [pad128(N)][pad32(K)], with input channels permuted where the
  HIP path keeps a concat buffer in a different (16-byte aligned) column order than the
  reference's torch.cat;
* LSTM gate rows regrouped per 16 hidden units (i,f,g,o of one unit adjacent) and both biases
  summed; biaffine weights transposed for the kernels; everything rounded to bf16 once.

All arithmetic here is fp32 on the host; nothing in this file runs per batch.
"""
import numpy as np
import torch


def p32(x):
    return (x + 31) // 32 * 32


def p128(x):
    return (x + 127) // 128 * 128


def _bf16(t):
    """fp32 -> Pair(hi, lo) of bf16 planes, t ~= hi + lo (the lo plane is used by bf16x3 only)."""
    from cube_amd.hip import Pair
    return Pair.split(t.to(torch.float32).contiguous())


def gate_column_order(half):
    """packed column n -> original conv output channel (or -1 for padding)."""
    nb = (half + 63) // 64
    order = np.full(nb * 128, -1, dtype=np.int64)
    for n in range(nb * 128):
        b, r = divmod(n, 128)
        wc, r = divmod(r, 64)
        part, j = divmod(r, 32)
        ch = b * 64 + wc * 32 + j
        if ch < half:
            order[n] = part * half + ch
    return order


# Weight-quantisation guards of the fixed-point kernels.  A layer runs in int8x2 / 12-bit fixed point only if the
# relative error its quantised weights put on a dot product with white inputs -- |w_q - w|_2 / |w|_2 of the worst
# output row -- stays below these bounds; otherwise the packer leaves the fixed-point operands out and the layer
# (for the gated stacks: the stack) runs in bf16x3.  Per-row scales make the error (max|w| / rms(w)) / (levels *
# sqrt(12)): 3.5e-5 for Gaussian rows at 32512 levels, 2.5e-4 for PyTorch's uniform LSTM init at 2032 levels (4-bit low slices; 6.2e-5 at the 8128 of 6-bit ones); a checkpoint with
# outlier weights 4x further out than that is refused instead of trusted.
I8_CONV_WEIGHT_ERR = 1.25e-4
LSTM8_WEIGHT_ERR = 1.0e-3


def quant_row_error(w, wq):
    """max over the non-zero rows of |wq - w|_2 / |w|_2 (w, wq: [rows, ...])."""
    w, wq = w.double().flatten(1), wq.double().flatten(1)
    den = w.norm(dim=1)
    ok = den > 0
    if not bool(ok.any()):
        return 0.0
    return float(((wq - w).norm(dim=1)[ok] / den[ok]).max())


def pack_gated_conv(sd, prefix, in_perm, double_first=0, bcast=None, int8=False, in_scale=None):
    """-> dict(W bf16 [N, taps*Kc], bias f32 [N], Kc, N, half).
    in_perm: for each column of the HIP activation buffer, the reference input channel (or -1).
    bcast = (reference channels, table [G, len(channels)]): input channels that carry the same
    vector table[g] at every position of a sequence (the language embedding the reference
    concatenates before each conv).  They are left out of the GEMM; their contribution goes into
    `bias_tab` f32 [G, 32, N]: entry [g, m] = bias + sum over the taps t in bit mask m of
    W[:, channels, t] @ table[g]  (m = the taps that fall inside the sequence at a position).
    `bias_tab_q` is the same from bf16-rounded weights and table (single-pass "bf16" contract).
    int8: also `W8` (Pair of int8 planes [N, taps*Kc], 64-channel chunks) and `w_scale` f32 [N] for
    cube_gemm_i8x2: w[n, :] = (256*hi + lo) * max|w[n, :]| / I8_ONE (include/cube_hip.h).
    in_scale (f32 per buffer column, with int8): the layer's int8x2 input holds x[:, c] / in_scale[c] (cube_quant_i8x2), so the
    fixed-point weights are those of W * in_scale; `W` (bf16) stays unscaled for the bf16x3 route."""
    w = sd[prefix + ".0.conv.weight"].float()  # [F, Cin, taps]
    b = sd[prefix + ".0.conv.bias"].float()
    s = sd[prefix + ".1.weight"].float() / torch.sqrt(sd[prefix + ".1.running_var"].float() + 1e-5)
    w = w * s[:, None, None]
    b = (b - sd[prefix + ".1.running_mean"].float()) * s + sd[prefix + ".1.bias"].float()
    F_, cin, taps = w.shape
    half = F_ // 2
    if double_first:
        w = w.clone()
        w[:, :double_first, :] *= 2.0
    Kc = p32(len(in_perm))
    order = gate_column_order(half)
    N = len(order)
    W = torch.zeros((N, taps, Kc), dtype=torch.float32)
    bias = torch.zeros(N, dtype=torch.float32)
    valid_n = torch.from_numpy(np.where(order >= 0)[0])
    src_n = torch.from_numpy(order[order >= 0])
    in_perm = np.asarray(in_perm)
    valid_k = torch.from_numpy(np.where(in_perm >= 0)[0])
    src_k = torch.from_numpy(in_perm[in_perm >= 0])
    sub = w[src_n][:, src_k, :]  # [n, k, taps]
    tmp = torch.zeros((len(valid_n), taps, Kc), dtype=torch.float32)
    tmp[:, :, valid_k] = sub.permute(0, 2, 1)
    W[valid_n] = tmp
    bias[valid_n] = b[src_n]
    q8 = None
    if int8 and (Kc % 64 == 0 or in_scale is not None):
        from cube_amd.hip import I8_ONE, I8_SHIFT, Pair
        Kc8 = (Kc + 63) // 64 * 64   # the int8x2 tile reads 64-channel chunks: a quantised input buffer is padded to that
        Ws = W
        if in_scale is not None:
            scv = torch.ones(Kc, dtype=torch.float32)
            scv[:len(in_scale)] = torch.as_tensor(np.asarray(in_scale), dtype=torch.float32)
            Ws = W * scv[None, None, :]
        wmax = Ws.abs().amax(dim=(1, 2)).double().clamp(min=1e-30)
        q = torch.round(Ws.double() / wmax[:, None, None] * I8_ONE).to(torch.int32)
        hi = (q + (1 << (I8_SHIFT - 1))) >> I8_SHIFT  # arithmetic shift = floor: hi in [-127, 127], lo in [-128, 127]
        lo = q - (hi << I8_SHIFT)
        assert int(hi.abs().max()) <= 127 and int(lo.min()) >= -128 and int(lo.max()) <= 127

        def chunks(t):
            if Kc8 != Kc:
                t = torch.cat([t, torch.zeros((N, taps, Kc8 - Kc), dtype=t.dtype)], dim=2)
            return t.view(N, taps, Kc8 // 64, 64).permute(0, 2, 1, 3).reshape(N, taps * Kc8).to(torch.int8).contiguous()
        q8 = dict(Kc8=Kc8, W8=Pair(chunks(hi), chunks(lo)), w_scale=(wmax * float(1 << I8_SHIFT) / (I8_ONE * I8_ONE)).float().contiguous())
        q_err = quant_row_error(Ws, q.double() * wmax[:, None, None] / I8_ONE)
        if q_err > I8_CONV_WEIGHT_ERR:  # outlier weights: this layer's stack stays in bf16x3
            q8 = dict(int8_refused=q_err)
    # K order of cube_gemm_bf16: 32-channel chunk outer, tap inner
    W = W.view(N, taps, Kc // 32, 32).permute(0, 2, 1, 3).reshape(N, taps * Kc)
    out = dict(W=_bf16(W), bias=bias.contiguous(), Kc=Kc, N=N, half=half, taps=taps)
    if q8 is not None:
        out.update(q8)
    if bcast is not None:
        chans, table = bcast
        chans = torch.as_tensor(np.asarray(chans), dtype=torch.long)
        assert taps <= 5
        for key, q in (("bias_tab", lambda t: t.double()), ("bias_tab_q", lambda t: t.to(torch.bfloat16).double())):
            wl = q(w[:, chans, :])                                        # [F, C, taps] (BN already folded in)
            contrib = torch.einsum("gc,fct->gtf", q(table.float()), wl)   # [G, taps, F]
            masks = torch.tensor([[(m >> t) & 1 for t in range(taps)] for m in range(32)], dtype=torch.float64)
            tab = torch.einsum("mt,gtf->gmf", masks, contrib) + b.double()[None, None, :]   # [G, 32, F]
            packed = torch.zeros((table.shape[0], 32, N), dtype=torch.float32)
            packed[:, :, valid_n] = tab[:, :, src_n].float()
            out[key] = packed.contiguous()
    return out


def pack_conv0_tables(sd, prefix, tok_emb, case_emb, lang_emb, max_bytes=64 << 20):
    """First gated char layer as table look-ups.  Its input at a position is an embedding look-up
    [tok_emb[c] | case_emb[k] | lang_emb[g]] (modules.py:451-455,470), so the Conv1d pre-activation
    is a sum of per-(tap, character, case) vectors:
      T[t][c*n_case + k][:] = W[:, tok channels, t] @ tok_emb[c] + W[:, case channels, t] @ case_emb[k]
    plus the (language, tap mask) bias table of `pack_gated_conv`.  Columns are laid out
    [tanh half | sigmoid half] in channel order.  fp64 -> fp32; `*_q` from bf16-rounded operands
    (single-pass contract).  Returns None when the tables would exceed max_bytes or half % 4 != 0."""
    w = sd[prefix + ".0.conv.weight"].float()  # [F, Cin, taps]
    b = sd[prefix + ".0.conv.bias"].float()
    s = sd[prefix + ".1.weight"].float() / torch.sqrt(sd[prefix + ".1.running_var"].float() + 1e-5)
    w = w * s[:, None, None]
    b = (b - sd[prefix + ".1.running_mean"].float()) * s + sd[prefix + ".1.bias"].float()
    F_, cin, taps = w.shape
    half = F_ // 2
    n_char, ce = tok_emb.shape
    n_case, ke = case_emb.shape
    if half % 4 or taps > 5 or taps * n_char * n_case * F_ * 4 > max_bytes:
        return None
    out = dict(taps=taps, half=half, n_case=n_case, n_combo=n_char * n_case)
    for sfx, q in (("", lambda t: t.double()), ("_q", lambda t: t.to(torch.bfloat16).double())):
        tt = torch.einsum("ce,fet->tcf", q(tok_emb.float()), q(w[:, :ce, :]))                 # [taps, n_char, F]
        tc = torch.einsum("ke,fet->tkf", q(case_emb.float()), q(w[:, ce:ce + ke, :]))          # [taps, n_case, F]
        T = (tt[:, :, None, :] + tc[:, None, :, :]).reshape(taps, n_char * n_case, F_)
        contrib = torch.einsum("gc,fct->gtf", q(lang_emb.float()), q(w[:, ce + ke:, :]))        # [G, taps, F]
        masks = torch.tensor([[(m >> t) & 1 for t in range(taps)] for m in range(32)], dtype=torch.float64)
        bt = torch.einsum("mt,gtf->gmf", masks, contrib) + b.double()[None, None, :]            # [G, 32, F]
        out["T" + sfx] = T.float().contiguous()
        out["btab" + sfx] = bt.float().contiguous()
    # the padding character (row 0 of both embeddings, zero under padding_idx = 0): if its table rows are exactly zero the
    # kernel need not read them (cube_char_conv0_lookup, zero_code)
    out["zero_code"] = 0 if all(float(out[k][:, 0, :].abs().max()) == 0.0 for k in ("T", "T_q")) else -1
    return out


def pack_linear(w, b, in_perm=None, out_rows=None, n_pad=None):
    """w [n_out, n_in] -> W bf16 [pad128(rows), pad32(cols)], bias f32.
    in_perm: buffer column -> reference input feature (-1 = zero column);
    out_rows: packed output column -> reference output feature (-1 = zero row) or ('one',)."""
    w = w.float()
    n_out, n_in = w.shape
    if in_perm is None:
        in_perm = np.arange(n_in)
    if out_rows is None:
        out_rows = np.arange(n_out)
    in_perm = np.asarray(in_perm)
    out_rows = np.asarray(out_rows)
    K = p32(len(in_perm))
    N = n_pad or p128(len(out_rows))
    W = torch.zeros((N, K), dtype=torch.float32)
    bias = torch.zeros(N, dtype=torch.float32)
    vn = np.where(out_rows >= 0)[0]
    vk = np.where(in_perm >= 0)[0]
    tmp = torch.zeros((len(vn), K), dtype=torch.float32)
    tmp[:, torch.from_numpy(vk)] = w[torch.from_numpy(out_rows[vn])][:, torch.from_numpy(in_perm[vk])]
    W[torch.from_numpy(vn)] = tmp
    if b is not None:
        bias[torch.from_numpy(vn)] = b.float()[torch.from_numpy(out_rows[vn])]
    return dict(W=_bf16(W), bias=bias.contiguous(), K=K, N=N, n_valid=len(out_rows))


def pack_table(t, width=None):
    """embedding table -> bf16 [rows, width] (width >= cols, zero padded)."""
    t = t.float()
    width = width or t.shape[1]
    out = torch.zeros((t.shape[0], width), dtype=torch.float32)
    out[:, :t.shape[1]] = t
    return _bf16(out)


class Layout:
    """Column layouts of the activation buffers (all offsets in bf16 elements)."""

    def __init__(self, cfg, n_ext, is_parser=True):
        self.char_emb, self.case_emb, self.lang = cfg.char_emb_size, 32, cfg.lang_emb_size
        self.char_half, self.word_half = cfg.char_filter_size // 2, cfg.cnn_filter // 2
        self.word_emb, self.P = cfg.word_emb_size, cfg.external_proj_size
        # XC0: [tok | case]; XC (layers >= 1): [g].  The language embedding the reference concatenates
        # to every position (modules.py:455,470) is folded into a per-(language, tap mask) bias table.
        self.xc0_w = p32(self.char_emb + self.case_emb)
        self.xc_w = p32(self.char_half)
        # XW0: [word | char | lang | ext]   (reference: word, char, ext, lang — parser.py:149-151)
        self.xw0_char = self.word_emb
        self.xw0_lang = self.word_emb + self.char_half
        self.xw0_ext = self.xw0_lang + self.lang
        self.xw0_n = self.xw0_ext + self.P
        self.xw0_w = p32(self.xw0_n)
        self.xw_w = p32(self.word_half)  # [g]; lang folded into the bias table of layers >= 1

    def xw0_perm(self):
        """buffer column -> reference channel of the first word-level conv / of `_r_emb`."""
        ref_word, ref_char = 0, self.word_emb
        ref_ext = ref_char + self.char_half
        ref_lang = ref_ext + self.P
        perm = list(range(ref_word, ref_word + self.word_emb)) + list(range(ref_char, ref_char + self.char_half))
        perm += list(range(ref_lang, ref_lang + self.lang)) + list(range(ref_ext, ref_ext + self.P))
        return np.asarray(perm)


def pack_wordgram(sd, prefix, n_layers, char_emb, case_emb, lang, half):
    """WordGram (modules.py:424-448): embedding tables, gated char convs (`x + skip` doubling folded into
    layers >= 1, language channels into bias tables), layer-0 look-up tables, `_pre_out`."""
    pk = {}
    pk["tok_emb"] = pack_table(sd[prefix + "._tok_emb.weight"])
    pk["case_emb"] = pack_table(sd[prefix + "._case_emb.weight"])
    pk["wn_lang_emb"] = pack_table(sd[prefix + "._lang_emb.weight"])
    pk["char_convs"] = []
    wn_lang = sd[prefix + "._lang_emb.weight"].float()
    for l in range(n_layers):
        cin = char_emb + case_emb if l == 0 else half  # then come the lang channels
        pk["char_convs"].append(pack_gated_conv(sd, "%s._convolutions_char.%d" % (prefix, l), np.arange(cin),
                                                double_first=half if l > 0 else 0,
                                                bcast=(cin + np.arange(lang), wn_lang), int8=l > 0))
    pk["char_l0"] = pack_conv0_tables(sd, prefix + "._convolutions_char.0", sd[prefix + "._tok_emb.weight"],
                                      sd[prefix + "._case_emb.weight"], wn_lang)
    pk["char_pre_out"] = pack_linear(sd[prefix + "._pre_out.linear_layer.weight"], sd[prefix + "._pre_out.linear_layer.bias"])
    return pk


def xw0_block_scales(sd, lay):
    """One quantisation scale per column of XW0 ([word | char | lang | ext], `Layout`): the largest |value| the column's
    block can hold.  The parser's learned root row (`_r_emb`, parser.py:153-156) is written into XW0 in buffer order."""
    blocks = [(0, lay.word_emb, float(sd["_word_emb.weight"].float().abs().max())),
              (lay.xw0_char, lay.xw0_char + lay.char_half, 1.0),
              (lay.xw0_lang, lay.xw0_lang + lay.lang, float(sd["_lang_emb.weight"].float().abs().max())),
              (lay.xw0_ext, lay.xw0_ext + lay.P, 1.0)]
    root = None
    if "_r_emb.weight" in sd:
        root = sd["_r_emb.weight"].float()[0][torch.from_numpy(lay.xw0_perm())].abs()
    sc = np.ones(lay.xw0_n, dtype=np.float32)
    for c0, c1, m in blocks:
        if root is not None:
            m = max(m, float(root[c0:c1].max()))
        # bf16 hi + lo of a value can exceed it by an ulp of the lo plane; a power-of-two margin keeps 1/scale exact enough
        sc[c0:c1] = max(m, 1e-30) * (1.0 + 2.0 ** -12)
    return sc


def pack_front(sd, cfg, enc_sizes, ext_dims, lay):
    """Pieces shared by Parser and Tagger: WordGram, ext MLPs, word-level conv stack, aux heads."""
    pk = pack_wordgram(sd, "_word_net", cfg.char_layers, lay.char_emb, lay.case_emb, lay.lang, lay.char_half)
    pk["ext"] = []
    for i, e in enumerate(ext_dims):
        h1 = pack_linear(sd["_ext_proj.%d._h1.linear_layer.weight" % i], sd["_ext_proj.%d._h1.linear_layer.bias" % i])
        hid = sd["_ext_proj.%d._h2.linear_layer.weight" % i].shape[1]
        h2 = pack_linear(sd["_ext_proj.%d._h2.linear_layer.weight" % i], sd["_ext_proj.%d._h2.linear_layer.bias" % i],
                         in_perm=np.concatenate([np.arange(hid), -np.ones(h1["N"] - hid, dtype=np.int64)]))
        pk["ext"].append(dict(h1=h1, h2=h2, E=e, Epad=p32(e)))
    if len(ext_dims) > 1 and len(set(ext_dims)) == 1 and len({ek["h1"]["N"] for ek in pk["ext"]}) == 1:
        # K LM layers of one width: their first (and their second) MLP layers as ONE batched launch each (cube_gemm_args.batch) -- stacked weights
        from cube_amd.hip import Pair
        for nm in ("h1", "h2"):  # (the second layers share one shape whenever the first ones do: hidden -> proj)
            hs = [ek[nm] for ek in pk["ext"]]
            pk["ext_%scat" % nm] = dict(W=Pair(torch.cat([h["W"].hi for h in hs], 0).contiguous(),
                                               torch.cat([h["W"].lo for h in hs], 0).contiguous()),
                                        bias=torch.cat([h["bias"] for h in hs]).contiguous(), K=hs[0]["K"], N=hs[0]["N"], n=len(hs))
    pk["word_emb"] = pack_table(sd["_word_emb.weight"])
    pk["lang_emb"] = pack_table(sd["_lang_emb.weight"])
    perm0 = lay.xw0_perm()
    pk["word_convs"] = []
    lang_tab = sd["_lang_emb.weight"].float()
    for l in range(cfg.cnn_layers):
        # layer 0 keeps its lang channels in the GEMM (the parser's root row carries its own, parser.py:153-156)
        if l == 0:
            # int8x2 for layer 0 as well: its input [word | char | lang | ext] is quantised with one scale per block -- the
            # largest magnitude the block can hold: the embedding tables' maxima, 1 for the two tanh outputs, and the root
            # row's own values where it has one (parser.py:153-156)
            sc = xw0_block_scales(sd, lay)
            cv = pack_gated_conv(sd, "_convs.%d" % l, perm0, int8=True, in_scale=sc)
            if "W8" in cv:
                inv = np.zeros(cv["Kc"], dtype=np.float32)   # padding columns: 0
                inv[:len(sc)] = 1.0 / sc
                cv["in_inv_scale"] = torch.from_numpy(inv)
            pk["word_convs"].append(cv)
        else:
            pk["word_convs"].append(pack_gated_conv(sd, "_convs.%d" % l, np.arange(lay.word_half),
                                                    bcast=(lay.word_half + np.arange(lay.lang), lang_tab), int8=True))
    # aux heads read [char_emb | lang], contiguous inside XW0
    n_u, n_x, n_a = enc_sizes["upos"], enc_sizes["xpos"], enc_sizes["attrs"]
    w = torch.cat([sd["_aupos.linear_layer.weight"], sd["_axpos.linear_layer.weight"], sd["_aattrs.linear_layer.weight"]], 0)
    b = torch.cat([sd["_aupos.linear_layer.bias"], sd["_axpos.linear_layer.bias"], sd["_aattrs.linear_layer.bias"]], 0)
    pk["aux"] = pack_linear(w, b)
    return pk


def lstm_gate_rows(H):
    """packed gate column (within one direction) -> reference row g*H + hid, or -1."""
    U = (H + 15) // 16
    rows = np.full(U * 64, -1, dtype=np.int64)
    for u in range(U):
        for g in range(4):
            for j in range(16):
                hid = u * 16 + j
                if hid < H:
                    rows[u * 64 + g * 16 + j] = g * H + hid
    return rows


def lstm8_lo_bits(H):
    """Bits of the low slice of W_hh in cube_bilstm_recurrent_i8: 6 when the packed slices of a direction fit the LDS of a
    CU beside the h images (H <= 208), else 4 (= cube_bilstm_lo_bits of include/cube_hip.h, restated for the packer)."""
    U = (H + 15) // 16
    KP = 32 * ((U + 1) // 2)
    n64 = KP // 64
    tailq = (16 * U - 64 * n64 + 7) // 8 if KP % 64 else 0
    lds6 = 4 * 16 * (KP + 16) + U * 4 * (n64 * 768 + tailq * 128) + U * 256
    return 6 if lds6 <= 160 * 1024 else 4


def pack_lstm_i8(whh_dirs, H, lo_bits=None):
    """Recurrent weights of one BiLSTM layer for cube_bilstm_recurrent_i8 (include/cube_hip.h).
    whh_dirs: two f32 [G, >= H] matrices in packed gate-row order (`lstm_gate_rows`; padding rows zero).
    Row n: q = round(w / max|w[n]| * 127 * 2^lo_bits) = 2^lo_bits * hi + lo, hi in [-127, 127] (int8), lo stored as the
    unsigned v = lo + 2^(lo_bits - 1).  K is padded to a multiple of 32: n64 = KP // 64 fragments of K = 64, then one of
    K = 32.  lo_bits: 6 (default when it fits the LDS, `lstm8_lo_bits`) or 4.
    -> None when the quantised rows miss LSTM8_WEIGHT_ERR, else
       dict(hi uint8 [2, U, 4, WHI_TILE], lo uint8 [2, U, 4, WLO_TILE], scale f32 [2, G], lo_bits) in MFMA-fragment order:
       hi fragment ks: [lane][16 bytes], byte j of lane (l15, lq) = hi[row u*64 + g*16 + l15][ks*64 + lq*16 + j];
       tail: [lane][8], k = n64*64 + lq*8 + j;
       lo, 4 bits: fragment [lane][8 bytes] = two words, word m byte b = v[k0 + 4*(2m) + b] | v[k0 + 4*(2m+1) + b] << 4
       with k0 = ks*64 + lq*16; tail: [lane][4], one word;
       lo, 6 bits: three words per lane, word m byte b = v[k0 + 4m + b] | ((v[k0 + 12 + b] >> 2m) & 3) << 6, stored as
       [lane][words 0, 1] then [lane][word 2] per fragment;
       tail: whole bytes [lane][8] for the first tailq groups of 16 lanes (the columns a hidden unit < 16 U can reach)."""
    from cube_amd.hip import LSTM8_HONE
    if lo_bits is None:
        lo_bits = lstm8_lo_bits(H)
    assert lo_bits in (4, 6)
    wstep = 1 << lo_bits
    wone = 127 * wstep
    G = whh_dirs[0].shape[0]
    U = G // 64
    KP = 32 * ((U + 1) // 2)
    n64, tail = KP // 64, (KP % 64) != 0
    tailq = (16 * U - 64 * n64 + 7) // 8 if tail else 0
    his, los, scales = [], [], []
    for w in whh_dirs:
        wp = torch.zeros((G, KP), dtype=torch.float64)
        wp[:, :H] = w[:, :H].double()
        wmax = wp.abs().amax(dim=1).clamp(min=1e-30)
        q = torch.round(wp / wmax[:, None] * wone).to(torch.int32)
        if quant_row_error(wp, q.double() * wmax[:, None] / wone) > LSTM8_WEIGHT_ERR * 16 / wstep:
            return None  # outlier weights: this layer keeps the streamed bf16x3 recurrence
        hi = (q + wstep // 2) >> lo_bits
        lo = q - (hi << lo_bits)
        assert int(hi.abs().max()) <= 127 and int(lo.min()) >= -(wstep // 2) and int(lo.max()) < wstep // 2
        v = (lo + wstep // 2).to(torch.uint8)
        hi8 = hi.to(torch.int8).view(torch.uint8)
        r = hi8.view(U, 4, 16, KP)                                   # [u][g][l15][k]
        nb = v.view(U, 4, 16, KP)
        hp, lp = [], []
        for ks in range(n64):
            f = r[:, :, :, ks * 64:(ks + 1) * 64].reshape(U, 4, 16, 4, 16)          # [u][g][l15][lq][j]
            hp.append(f.permute(0, 1, 3, 2, 4).reshape(U, 4, 1024))                  # lane = lq*16 + l15
            n_ = nb[:, :, :, ks * 64:(ks + 1) * 64].reshape(U, 4, 16, 4, 4, 4)       # [u][g][l15][lq][reg m][byte b]
            if lo_bits == 4:
                pk = (n_[..., 0::2, :] | (n_[..., 1::2, :] << 4))                    # [u][g][l15][lq][2][4]
                lp.append(pk.permute(0, 1, 3, 2, 4, 5).reshape(U, 4, 512))
            else:
                top = n_[..., 3, :]                                                  # the values 12 + b: two bits into each word
                pk = torch.stack([n_[..., m, :] | (((top >> (2 * m)) & 3) << 6) for m in range(3)], dim=-2)  # [..][lq][3][4]
                pk = pk.permute(0, 1, 3, 2, 4, 5)                                    # [u][g][lq][l15][3][4]
                lp.append(torch.cat([pk[..., :2, :].reshape(U, 4, 512), pk[..., 2, :].reshape(U, 4, 256)], dim=2))
        if tail:
            f = r[:, :, :, n64 * 64:n64 * 64 + 32].reshape(U, 4, 16, 4, 8)
            hp.append(f.permute(0, 1, 3, 2, 4).reshape(U, 4, 512))
            if lo_bits == 4:
                n_ = nb[:, :, :, n64 * 64:n64 * 64 + 32].reshape(U, 4, 16, 4, 2, 4)  # [..][lq][reg m][byte b]
                pk = (n_[..., 0, :] | (n_[..., 1, :] << 4))                          # [u][g][l15][lq][4]
                lp.append(pk.permute(0, 1, 3, 2, 4).reshape(U, 4, 256))
            else:
                n_ = nb[:, :, :, n64 * 64:n64 * 64 + 8 * tailq].reshape(U, 4, 16, tailq, 8)   # [u][g][l15][lq][j]
                lp.append(n_.permute(0, 1, 3, 2, 4).reshape(U, 4, tailq * 128))
        his.append(torch.cat(hp, dim=2))
        los.append(torch.cat(lp, dim=2))
        scales.append((wmax / (LSTM8_HONE * 127.0)).float())
    return dict(hi=torch.stack(his).contiguous(), lo=torch.stack(los).contiguous(), scale=torch.stack(scales).contiguous(),
                lo_bits=lo_bits)


def pack_parser(sd, cfg, enc_sizes, ext_dims):
    """-> (dict of packed CPU tensors / ints, Layout)."""
    lay = Layout(cfg, len(ext_dims))
    pk = pack_front(sd, cfg, enc_sizes, ext_dims, lay)
    half, L_, P, H = lay.word_half, lay.lang, lay.P, cfg.rnn_size
    n_u, n_x, n_a, n_l = enc_sizes["upos"], enc_sizes["xpos"], enc_sizes["attrs"], enc_sizes["labels"]
    pk["r_emb"] = pack_table(sd["_r_emb.weight"][:, torch.from_numpy(lay.xw0_perm())], lay.xw0_w)
    # morphology: XM = [hidden | lang];  PM = [pre_morpho | lang | upos_emb]
    pk["pre_morpho"] = pack_linear(sd["_pre_morpho.linear_layer.weight"], sd["_pre_morpho.linear_layer.bias"])
    pk["upos"] = pack_linear(sd["_upos.linear_layer.weight"], sd["_upos.linear_layer.bias"])
    w = torch.cat([sd["_attrs.linear_layer.weight"], sd["_xpos.linear_layer.weight"]], 0)
    b = torch.cat([sd["_attrs.linear_layer.bias"], sd["_xpos.linear_layer.bias"]], 0)
    pk["attrs_xpos"] = pack_linear(w, b)
    pk["upos_emb"] = pack_table(sd["_upos_emb.weight"])
    # LSTM.  XL0 = [x_parse | lang | ext] (reference: x_parse, ext, lang — parser.py:195-200)
    rows1 = lstm_gate_rows(H)
    G = len(rows1)
    pk["lstm"] = []
    for l in range(cfg.rnn_layers):
        if l == 0:
            in_perm = np.concatenate([np.arange(half), half + P + np.arange(L_), half + np.arange(P)])
        else:
            in_perm = np.arange(2 * H)
        ws, bs, hh = [], [], []
        for suffix in ("", "_reverse"):
            ws.append(sd["_rnn.weight_ih_l%d%s" % (l, suffix)].float())
            bs.append(sd["_rnn.bias_ih_l%d%s" % (l, suffix)].float() + sd["_rnn.bias_hh_l%d%s" % (l, suffix)].float())
            whh = sd["_rnn.weight_hh_l%d%s" % (l, suffix)].float()
            KH = p32(H)
            wp = torch.zeros((G, KH), dtype=torch.float32)
            v = np.where(rows1 >= 0)[0]
            wp[torch.from_numpy(v), :H] = whh[torch.from_numpy(rows1[v])]
            hh.append(wp)
        out_rows = np.concatenate([rows1, np.where(rows1 >= 0, rows1 + 4 * H, -1)])
        ih = pack_linear(torch.cat(ws, 0), torch.cat(bs, 0), in_perm=in_perm, out_rows=out_rows)
        # recurrent weights in MFMA-fragment order [dir][u][gate][k-step][lane][8] (cube_bilstm_recurrent)
        U, KS = G // 64, KH // 32
        w4 = torch.stack(hh, 0).view(2, U, 4, 16, KS, 4, 8)       # [d][u][g][l15][ks][lq][j]
        frag = w4.permute(0, 1, 2, 4, 5, 3, 6).reshape(2, U, 4, KS, 64, 8)  # lane = lq*16 + l15
        pk["lstm"].append(dict(ih=ih, whh=_bf16(frag.contiguous()), G=G, whh8=pack_lstm_i8(hh, H)))
    # _pre_out reads [h_fwd | h_bwd | lang]  (reference order)
    pk["pre_out"] = pack_linear(sd["_pre_out.linear_layer.weight"], sd["_pre_out.linear_layer.bias"])
    pp_w = pk["pre_out"]["N"]  # width of the pre_parsing buffer (pad columns are tanh(0) = 0)
    pp_perm = np.concatenate([np.arange(cfg.pre_parser_size), -np.ones(pp_w - cfg.pre_parser_size, dtype=np.int64)])

    def biaffine_rows(name, hs):
        D = p32(hs + 1)
        w = torch.cat([sd[name + ".W1.weight"], sd[name + ".W2.weight"]], 0).float()
        b = torch.cat([sd[name + ".W1.bias"], sd[name + ".W2.bias"]], 0).float()
        out_rows = -np.ones(2 * D, dtype=np.int64)
        out_rows[:hs] = np.arange(hs)
        out_rows[D:D + hs] = hs + np.arange(hs)
        return D, w, b, out_rows

    def biaffine_in(name, hs):
        """[h1 | 1 | 0.. | h2 | 1 | 0..] with D = pad32(hs+1) columns each; ones via relu(0*x + 1)."""
        D, w, b, out_rows = biaffine_rows(name, hs)
        p = pack_linear(w, b, in_perm=pp_perm, out_rows=out_rows)
        p["bias"][hs] = 1.0
        p["bias"][D + hs] = 1.0
        p["D"] = D
        return p

    hs, ls = cfg.head_size, cfg.label_size
    pk["head_in"] = biaffine_in("_head", hs)
    Dh = pk["head_in"]["D"]
    wb = sd["_head.scorer.W_bilin.weight"].float()[:, :, 0]  # [p, q]
    wpad = torch.zeros((Dh, Dh), dtype=torch.float32)
    wpad[:hs + 1, :hs + 1] = wb.t()  # packed [q][p]
    pk["head_bilin"] = dict(W=_bf16(torch.cat([wpad, torch.zeros((p128(Dh) - Dh, Dh))], 0)), N=p128(Dh), K=Dh)
    pk["label_in"] = biaffine_in("_label", ls)
    Dl = pk["label_in"]["D"]
    wl = sd["_label.scorer.W_bilin.weight"].float()  # [o, p, q]
    wlp = torch.zeros((n_l, Dl, Dl), dtype=torch.float32)
    wlp[:, :ls + 1, :ls + 1] = wl.permute(0, 2, 1)  # [o][q][p]
    pk["label_bilin"] = dict(W=_bf16(wlp), bias=sd["_label.scorer.W_bilin.bias"].float().contiguous(), D=Dl)
    # both biaffine input projections read pre_parsing: one GEMM -> [h1;1 | h2;1 | x1;1 | x2;1]
    # (`_get_labels` on pre_parsing handed in by the caller keeps the separate label_in)
    _, wh_, bh_, rh_ = biaffine_rows("_head", hs)
    _, wl_, bl_, rl_ = biaffine_rows("_label", ls)
    rows = np.concatenate([rh_, np.where(rl_ >= 0, rl_ + wh_.shape[0], -1)])
    p = pack_linear(torch.cat([wh_, wl_], 0), torch.cat([bh_, bl_], 0), in_perm=pp_perm, out_rows=rows)
    for c in (hs, Dh + hs, 2 * Dh + ls, 2 * Dh + Dl + ls):
        p["bias"][c] = 1.0
    p["Dh"], p["Dl"] = Dh, Dl
    pk["bi_in"] = p
    return pk, lay


def pack_tagger(sd, cfg, enc_sizes, ext_dims):
    lay = Layout(cfg, len(ext_dims), is_parser=False)
    pk = pack_front(sd, cfg, enc_sizes, ext_dims, lay)
    half, L_ = lay.word_half, lay.lang
    # TU = [x+res | lang | upos_emb]; reference: upos reads [x, lang], xpos/attrs read
    # [upos_emb, lang, x]  (tagger.py:148-161)
    pk["upos"] = pack_linear(sd["_upos.linear_layer.weight"], sd["_upos.linear_layer.bias"])
    perm = np.concatenate([64 + L_ + np.arange(half), 64 + np.arange(L_), np.arange(64)])
    w = torch.cat([sd["_attrs.linear_layer.weight"], sd["_xpos.linear_layer.weight"]], 0)
    b = torch.cat([sd["_attrs.linear_layer.bias"], sd["_xpos.linear_layer.bias"]], 0)
    pk["attrs_xpos"] = pack_linear(w, b, in_perm=perm)
    pk["upos_emb"] = pack_table(sd["_upos_emb.weight"])
    return pk, lay


def to_device(obj, device, with_lo=True):
    """Upload; bf16x3 (with_lo) keeps both planes and the exact bias tables, single-pass bf16 keeps
    the hi plane and the tables built from bf16-rounded operands."""
    from cube_amd.hip import Pair
    if isinstance(obj, Pair):
        return Pair(obj.hi.to(device), obj.lo.to(device) if (with_lo and obj.lo is not None) else None)
    if isinstance(obj, torch.Tensor):
        return obj.to(device)
    if isinstance(obj, dict):
        out = {k: to_device(v, device, with_lo) for k, v in obj.items() if k not in ("bias_tab_q", "T_q", "btab_q")}
        for k in ("bias_tab", "T", "btab"):  # single-pass bf16: the tables built from bf16-rounded operands
            if k + "_q" in obj and not with_lo:
                out[k] = obj[k + "_q"].to(device)
        return out
    if isinstance(obj, list):
        return [to_device(v, device, with_lo) for v in obj]
    return obj


# ------------------------------------------------------------------------------------------
# packed weights as a file, for hosts that cannot run this packer (include/cube_hip.h: cube_packfile_*)
# ------------------------------------------------------------------------------------------
PACK_DTYPES = {torch.float32: 0, torch.bfloat16: 1, torch.int8: 2, torch.int32: 3, torch.int64: 4, torch.uint8: 5}


def flatten_packed(pk, layout=None):
    """The packer's nested dict -> [(name, tensor | int)]: keys joined by dots, list indices as numbers, `hi` / `lo` for
    the planes of a Pair, `layout.<attr>` for the integer attributes of the layout object.  Floats, strings, None and
    other Python-side bookkeeping are left out (a non-Python host has no use for them)."""
    from cube_amd.hip import Pair
    out = []

    def walk(prefix, v):
        if isinstance(v, Pair):
            walk(prefix + ".hi", v.hi)
            if v.lo is not None:
                walk(prefix + ".lo", v.lo)
        elif isinstance(v, torch.Tensor):
            if v.dtype not in PACK_DTYPES or v.dim() > 8:   # never drop an operand silently
                raise TypeError("save_packed: %s has type %s / %d dimensions, which the file format does not hold"
                                % (prefix, v.dtype, v.dim()))
            out.append((prefix, v.detach().cpu().contiguous()))
        elif isinstance(v, bool):
            out.append((prefix, int(v)))
        elif isinstance(v, (int, np.integer)):
            out.append((prefix, int(v)))
        elif isinstance(v, dict):
            for k in v:
                walk(prefix + "." + str(k) if prefix else str(k), v[k])
        elif isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                walk(prefix + "." + str(i) if prefix else str(i), x)

    walk("", pk)
    if layout is not None:
        for k, v in sorted(vars(layout).items()):
            if isinstance(v, (int, np.integer)) and not isinstance(v, bool):
                out.append(("layout." + k, int(v)))
    return out


def save_packed(pk, path, layout=None):
    """Write the packed weights (host copy, both planes of every pair) to `path` in the format that cube_packfile_open
    reads (nlp-cube_amd/csrc/packfile.hip).  -> number of entries."""
    import struct
    items = flatten_packed(pk, layout)
    table, blobs, off = [], [], 0
    for name, v in items:
        nb = name.encode("utf-8")
        if isinstance(v, int):
            raw, code, dims = struct.pack("<q", v), 4, ()
        else:
            code, dims = PACK_DTYPES[v.dtype], tuple(v.shape)
            raw = (v.view(torch.int16) if v.dtype == torch.bfloat16 else v).numpy().tobytes()
        table.append(struct.pack("<H", len(nb)) + nb + struct.pack("<BB", code, len(dims)) +
                     b"".join(struct.pack("<q", d) for d in dims) + struct.pack("<QQ", off, len(raw)))
        blobs.append((off, raw))
        off = (off + len(raw) + 255) // 256 * 256
    tab = b"".join(table)
    data_off = (32 + len(tab) + 255) // 256 * 256
    with open(path, "wb") as f:
        f.write(b"CUBEPACK" + struct.pack("<IIQQ", 1, len(items), len(tab), data_off))
        f.write(tab)
        for o, raw in blobs:
            f.seek(data_off + o)
            f.write(raw)
        f.truncate(data_off + off if off else data_off)
    return len(items)


# ------------------------------------------------------------------------------------------
# lemmatizer (reference: cube/networks/lemmatizer.py:24-69)
# ------------------------------------------------------------------------------------------
def p8(x):
    return (x + 7) // 8 * 8


def _dense(rows, cols):
    return torch.zeros((rows, cols), dtype=torch.float32)


def pack_conv_plain(w, b, in_perm):
    """Conv1d weight [F, Cin, taps] -> W bf16 [pad128(F), taps*Kc] in the K order of cube_gemm_bf16
    (32-channel chunk outer, tap inner), bias f32 [pad128(F)].  in_perm: buffer column -> input channel."""
    w = w.float()
    F_, cin, taps = w.shape
    in_perm = np.asarray(in_perm)
    Kc, N = p32(len(in_perm)), p128(F_)
    vk = np.where(in_perm >= 0)[0]
    W = torch.zeros((N, taps, Kc), dtype=torch.float32)
    W[:F_, :, torch.from_numpy(vk)] = w[:, torch.from_numpy(in_perm[vk]), :].permute(0, 2, 1)
    W = W.view(N, taps, Kc // 32, 32).permute(0, 2, 1, 3).reshape(N, taps * Kc)
    bias = torch.zeros(N, dtype=torch.float32)
    bias[:F_] = b.float()
    return dict(W=_bf16(W), bias=bias, Kc=Kc, N=N, taps=taps)


class LemmaLayout:
    """Column layouts of the lemmatizer's activation buffers (bf16 elements; segments padded to 8
    so that every row copy is 16-byte aligned).
      XE0 [char | case | upos]                      input of the first gated conv (lang: bias table)
      ENC [x+res 256 | case | upos | lang]          encoder_output (lemmatizer.py:124), C = enc_c
      XD  [ctx (ENC layout) | prev char | h0 | h1 | c1]   decoder state row of a word"""

    def __init__(self, cfg, n_chars):
        self.ce, self.ue, self.le, self.D, self.P = (cfg.char_emb_size, cfg.upos_emb_size, cfg.lang_emb_size,
                                                     cfg.decoder_size, cfg.att_proj_size)
        self.half, self.case = 256, 16
        self.n_char, self.n_case = n_chars + 2, 4
        self.xe0_case, self.xe0_upos = p8(self.ce), p8(self.ce) + self.case
        self.xe0_w = p32(self.xe0_upos + p8(self.ue))
        self.enc_case, self.enc_upos = self.half, self.half + self.case
        self.enc_lang = self.enc_upos + p8(self.ue)
        self.enc_c = self.enc_lang + p8(self.le)
        self.enc_w = p32(self.enc_c)
        self.xd_emb = self.enc_w
        self.xd_h0 = self.xd_emb + p32(self.ce)
        self.xd_h1 = self.xd_h0 + p32(self.D)
        self.xd_c1 = self.xd_h1 + p32(self.D)
        self.xd_w = self.xd_c1 + p32(self.D)
        # Q (f32, per step): [5 tap projections of the query, ld_tap each, pad to 128 | char logits | case logits]
        self.ld_tap = p8(self.P)
        self.q_char = p128(5 * self.ld_tap)
        self.q_case = self.q_char + self.n_char
        self.q_w = self.q_char + p128(self.n_char + self.n_case)

    def xe0_perm(self):
        """XE0 column -> input channel of the first conv ([char, case, upos, lang], lemmatizer.py:92-98,113)."""
        perm = -np.ones(self.xe0_w, dtype=np.int64)
        perm[:self.ce] = np.arange(self.ce)
        perm[self.xe0_case:self.xe0_case + self.case] = self.ce + np.arange(self.case)
        perm[self.xe0_upos:self.xe0_upos + self.ue] = self.ce + self.case + np.arange(self.ue)
        return perm

    def enc_perm(self):
        """ENC column -> channel of encoder_output ([x, case, upos, lang], lemmatizer.py:124)."""
        perm = -np.ones(self.enc_w, dtype=np.int64)
        perm[:self.half + self.case] = np.arange(self.half + self.case)
        perm[self.enc_upos:self.enc_upos + self.ue] = self.half + self.case + np.arange(self.ue)
        perm[self.enc_lang:self.enc_lang + self.le] = self.half + self.case + self.ue + np.arange(self.le)
        return perm


def lstm_epilogue_rows(units):
    """Packed gate row -> row of torch's [4*units] gate order (i, f, g, o), -1 = padding, for CUBE_EPI_LSTM:
    column ub*64 + part*32 + 2*j + odd holds gate 2*part + odd of unit ub*16 + j (include/cube_hip.h)."""
    nb = (units + 15) // 16
    perm = -np.ones(nb * 64, dtype=np.int64)
    for u in range(units):
        ub, j = divmod(u, 16)
        for gate in range(4):
            perm[ub * 64 + (gate >> 1) * 32 + 2 * j + (gate & 1)] = gate * units + u
    return perm


def pack_lemmatizer(sd, cfg, n_chars):
    lay = LemmaLayout(cfg, n_chars)
    ce, ue, le, D, P, half = lay.ce, lay.ue, lay.le, lay.D, lay.P, lay.half
    pk = {}
    pk["char_emb"] = pack_table(sd["_char_emb.weight"], p8(ce))
    pk["case_emb"] = pack_table(sd["_case_emb.weight"])
    pk["upos_emb"] = pack_table(sd["_upos_emb.weight"], p8(ue))
    pk["lang_emb"] = pack_table(sd["_lang_emb.weight"], p8(le))
    lang = sd["_lang_emb.weight"].float()
    pk["convs"] = []
    n_layers = sum(1 for k in sd if k.startswith("_convolutions_char.") and k.endswith(".0.conv.weight"))
    for l in range(n_layers):
        if l == 0:
            cv = pack_gated_conv(sd, "_convolutions_char.0", lay.xe0_perm(), bcast=(ce + lay.case + ue + np.arange(le), lang))
        else:  # x = dropout(tmp) + skip = 2 tmp (lemmatizer.py:110-111,121): the factor goes into the weights
            cv = pack_gated_conv(sd, "_convolutions_char.%d" % l, np.arange(half), double_first=half,
                                 bcast=(half + np.arange(le), lang))
        pk["convs"].append(cv)
    # attention Conv1d over [query (D) | encoder_output]: the encoder part once per batch ...
    wa, ba = sd["_attention.attn.conv.weight"].float(), sd["_attention.attn.conv.bias"].float()
    enc_perm = lay.enc_perm()
    pk["att_enc"] = pack_conv_plain(wa, ba, np.where(enc_perm >= 0, enc_perm + D, -1))
    pk["att_v"] = sd["_attention.v"].float().contiguous()
    # ... the query part per step: 5 tap projections of the query (= cell state c1, lemmatizer.py:144)
    KD = p32(D)
    Wt = _dense(lay.q_char, KD)
    for k in range(wa.shape[2]):
        Wt[k * lay.ld_tap:k * lay.ld_tap + P, :D] = wa[:, :D, k]
    pk["query"] = dict(W=_bf16(Wt), bias=torch.zeros(lay.q_char, dtype=torch.float32), K=KD, N=lay.q_char)
    # the two output heads read h1
    nh = lay.q_w - lay.q_char
    Wh, bh = _dense(nh, KD), torch.zeros(nh, dtype=torch.float32)
    Wh[:lay.n_char, :D] = sd["_output_char.linear_layer.weight"].float()
    bh[:lay.n_char] = sd["_output_char.linear_layer.bias"].float()
    Wh[lay.n_char:lay.n_char + lay.n_case, :D] = sd["_output_case.linear_layer.weight"].float()
    bh[lay.n_char:lay.n_char + lay.n_case] = sd["_output_case.linear_layer.bias"].float()
    pk["heads"] = dict(W=_bf16(Wh), bias=bh, K=KD, N=nh)
    # decoder LSTM: layer 0 reads [ctx | prev char | h0], layer 1 reads [h0 | h1]
    ng = (4 * D + 255) // 256 * 256   # multiples of 256 columns: large batches run on the 256x256 tiles
    vk = torch.from_numpy(np.where(enc_perm >= 0)[0])
    src = torch.from_numpy(enc_perm[enc_perm >= 0])
    w_ih, w_hh = sd["_decoder.weight_ih_l0"].float(), sd["_decoder.weight_hh_l0"].float()
    W0 = _dense(ng, lay.xd_h1)
    tmp = torch.zeros((4 * D, lay.enc_w), dtype=torch.float32)
    tmp[:, vk] = w_ih[:, src]
    W0[:4 * D, :lay.enc_w] = tmp
    n_ctx = half + lay.case + ue + le
    W0[:4 * D, lay.xd_emb:lay.xd_emb + ce] = w_ih[:, n_ctx:n_ctx + ce]
    W0[:4 * D, lay.xd_h0:lay.xd_h0 + D] = w_hh
    b0 = torch.zeros(ng, dtype=torch.float32)
    b0[:4 * D] = sd["_decoder.bias_ih_l0"].float() + sd["_decoder.bias_hh_l0"].float()
    pk["lstm0"] = dict(W=_bf16(W0), bias=b0, K=lay.xd_h1, N=ng)
    W1 = _dense(ng, 2 * KD)
    W1[:4 * D, :D] = sd["_decoder.weight_ih_l1"].float()
    W1[:4 * D, KD:KD + D] = sd["_decoder.weight_hh_l1"].float()
    b1 = torch.zeros(ng, dtype=torch.float32)
    b1[:4 * D] = sd["_decoder.bias_ih_l1"].float() + sd["_decoder.bias_hh_l1"].float()
    pk["lstm1"] = dict(W=_bf16(W1), bias=b1, K=2 * KD, N=ng)
    # the same two layers with the gate rows in the order of CUBE_EPI_LSTM (include/cube_hip.h): the four pre-activations
    # of a hidden unit in two neighbouring lanes of the GEMM's accumulator tile, so that the cell update runs in the epilogue
    perm = lstm_epilogue_rows(D)
    ngf = (len(perm) + 255) // 256 * 256
    for name, Wf, bf in (("lstm0f", W0, b0), ("lstm1f", W1, b1)):
        Wp, bp = _dense(ngf, Wf.shape[1]), torch.zeros(ngf, dtype=torch.float32)
        ok = torch.from_numpy(perm >= 0)
        srcr = torch.from_numpy(perm[perm >= 0])
        Wp[:len(perm)][ok] = Wf[srcr]
        bp[:len(perm)][ok] = bf[srcr]
        pk[name] = dict(W=_bf16(Wp), bias=bp, K=Wf.shape[1], N=ngf)
    # start frame (lemmatizer.py:128-131) laid out as the [ctx | prev char] part of a decoder row
    sf = sd["_start_frame.weight"].float()[0]
    row = torch.zeros((1, lay.xd_h0), dtype=torch.float32)
    row[0, vk] = sf[src]
    row[0, lay.xd_emb:lay.xd_emb + ce] = sf[n_ctx:n_ctx + ce]
    pk["start"] = _bf16(row)
    return pk, lay


# ------------------------------------------------------------------------------------------
# tokenizer (reference: cube/networks/tokenizer.py:26-59)
# ------------------------------------------------------------------------------------------
class TokenizerLayout:
    """XW0 [ext | char 256 | spa 16] (language: bias table); FIN [x+res | lang] (input of `_output`)."""

    def __init__(self, cfg):
        # WordGram keeps its class defaults (modules.py:425-426)
        self.char_emb, self.case_emb, self.lang, self.char_half = 256, 32, 32, 256
        self.xc0_w = p32(self.char_emb + self.case_emb)
        self.xc_w = p32(self.char_half)
        self.P, self.word_half, self.word_lang, self.spa = cfg.external_proj_size, cfg.cnn_filter // 2, cfg.lang_emb_size, 16
        self.xw0_char = p8(self.P)
        self.xw0_spa = self.xw0_char + self.char_half
        self.xw0_w = p32(self.xw0_spa + self.spa)
        self.xw_w = p32(self.word_half)
        self.fin_w = p32(self.word_half + p8(self.word_lang))

    def xw0_perm(self):
        """XW0 column -> input channel of the first conv ([ext, char, spa, lang], tokenizer.py:100-102)."""
        perm = -np.ones(self.xw0_w, dtype=np.int64)
        perm[:self.P] = np.arange(self.P)
        perm[self.xw0_char:self.xw0_char + self.char_half] = self.P + np.arange(self.char_half)
        perm[self.xw0_spa:self.xw0_spa + self.spa] = self.P + self.char_half + np.arange(self.spa)
        return perm


def pack_tokenizer(sd, cfg, enc_sizes, ext_dims):
    lay = TokenizerLayout(cfg)
    pk = pack_wordgram(sd, "_wg", 3, lay.char_emb, lay.case_emb, lay.lang, lay.char_half)
    pk["ext"] = []
    for i, e in enumerate(ext_dims):
        h1 = pack_linear(sd["_ext_proj.%d._h1.linear_layer.weight" % i], sd["_ext_proj.%d._h1.linear_layer.bias" % i])
        hid = sd["_ext_proj.%d._h2.linear_layer.weight" % i].shape[1]
        h2 = pack_linear(sd["_ext_proj.%d._h2.linear_layer.weight" % i], sd["_ext_proj.%d._h2.linear_layer.bias" % i],
                         in_perm=np.concatenate([np.arange(hid), -np.ones(h1["N"] - hid, dtype=np.int64)]))
        pk["ext"].append(dict(h1=h1, h2=h2, E=e, Epad=p32(e)))
    pk["spa_emb"] = pack_table(sd["_spa_emb.weight"])
    pk["lang_emb"] = pack_table(sd["_lang_emb.weight"], p8(lay.word_lang))
    lang_tab = sd["_lang_emb.weight"].float()
    first = lay.P + lay.char_half + lay.spa
    pk["word_convs"] = []
    for l in range(cfg.cnn_layers):
        if l == 0:
            pk["word_convs"].append(pack_gated_conv(sd, "_convs.0", lay.xw0_perm(), bcast=(first + np.arange(lay.word_lang), lang_tab)))
        else:
            pk["word_convs"].append(pack_gated_conv(sd, "_convs.%d" % l, np.arange(lay.word_half),
                                                    bcast=(lay.word_half + np.arange(lay.word_lang), lang_tab)))
    perm = -np.ones(lay.fin_w, dtype=np.int64)
    perm[:lay.word_half + lay.word_lang] = np.arange(lay.word_half + lay.word_lang)
    pk["output"] = pack_linear(sd["_output.linear_layer.weight"], sd["_output.linear_layer.bias"], in_perm=perm)
    return pk, lay
