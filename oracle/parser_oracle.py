"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in PyTorch-CPU fp32 + NumPy, of the reference's batched
tagger/parser inference path:

  MorphoCollate.collate_fn   cube/networks/utils.py:226-330
  unpack                     cube/networks/utils.py:17-32
  WordGram.forward           cube/networks/modules.py:450-484
  Parser.forward             cube/networks/parser.py:109-216
  Parser._get_labels         cube/networks/parser.py:218-233 (+ modules.py:536-547)
  Parser.process             cube/networks/parser.py:404-448
  Tagger.forward / process   cube/networks/tagger.py:91-162, 288-329
  PairwiseBilinear / DeepBiaffine   cube/networks/modules.py:503-579

Two arithmetic modes:
  quant="fp32"  the reference's arithmetic (separate conv + eval BatchNorm, fp32 everywhere).
                Pinned against the reference itself by tests/golden/*.npz (generated in the
                build container by tests/golden/make_golden.py with the reference imported).
  quant="bf16"  the arithmetic contract of the HIP path (DESIGN.md "quantisation points"):
                BatchNorm folded into the conv weights, every GEMM operand (weights and
                activations) rounded to bf16 (round-to-nearest-even), fp32 accumulation,
                fp32 biases / gates / residual sums / cell state.  This is what the -m gpu
                parity tests compare the kernels with (north_star: "within 1e-3 relative in
                fp32-accumulated bf16").

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import cle_oracle


# ------------------------------------------------------------------------------------------
# featurisation  (reference: utils.py:226-330)
# ------------------------------------------------------------------------------------------
def collate(batch, enc):
    """Literal restatement of MorphoCollate.collate_fn for inference (y_* targets other than
    shapes are not produced; `process` deletes/ignores them, parser.py:414)."""
    sent_len = np.array([len(s.words) for s in batch], dtype=np.int64)
    word_len = np.array([len(w.word) for s in batch for w in s.words], dtype=np.int64)
    B, L, N, W = len(batch), int(sent_len.max()), len(word_len), int(word_len.max())
    K = len(batch[0].words[0].emb)
    x_sent = np.zeros((B, L), dtype=np.int64)
    x_sent_masks = np.zeros((B, L), dtype=np.float64)
    x_word = np.zeros((N, W), dtype=np.int64)
    x_word_case = np.zeros((N, W), dtype=np.int64)
    x_word_masks = np.zeros((N, W), dtype=np.float64)
    x_lang_sent = np.zeros(B, dtype=np.int64)
    x_lang_word = np.zeros(N, dtype=np.int64)
    embs = [[] for _ in range(K)]
    unk_w, unk_c = enc.word2int["<UNK>"], enc.char2int["<UNK>"]
    n = 0
    for b, sent in enumerate(batch):
        x_lang_sent[b] = sent.lang_id + 1
        for i, w in enumerate(sent.words):
            for k in range(K):
                embs[k].append(w.emb[k])
            x_sent_masks[b, i] = 1
            x_sent[b, i] = enc.word2int.get(w.word.lower(), unk_w)
            x_lang_word[n] = sent.lang_id + 1
            for p, ch in enumerate(w.word):
                x_word_masks[n, p] = 1
                if ch.lower() == ch.upper():
                    x_word_case[n, p] = 1
                elif ch.lower() != ch:
                    x_word_case[n, p] = 2
                else:
                    x_word_case[n, p] = 3
                x_word[n, p] = enc.char2int.get(ch.lower(), unk_c)
            n += 1
    return {
        "x_sent": torch.from_numpy(x_sent), "x_lang_sent": torch.from_numpy(x_lang_sent),
        "x_word": torch.from_numpy(x_word), "x_word_case": torch.from_numpy(x_word_case),
        "x_lang_word": torch.from_numpy(x_lang_word), "x_sent_len": torch.from_numpy(sent_len),
        "x_word_len": torch.from_numpy(word_len), "x_sent_masks": torch.from_numpy(x_sent_masks),
        "x_word_masks": torch.from_numpy(x_word_masks),
        "x_word_embeddings": [torch.tensor(np.asarray(e), dtype=torch.float32) for e in embs],
    }


def unpack(data, sizes, max_size):
    """reference: utils.py:17-32 — packed [N, D] -> padded [B, max_size, D], zeros at pads."""
    out = torch.zeros((len(sizes), max_size, data.shape[-1]), dtype=torch.float32)
    pos = 0
    for b, n in enumerate(sizes):
        n = int(n)
        out[b, :n] = data[pos:pos + n]
        pos += n
    return out


# ------------------------------------------------------------------------------------------
# arithmetic helpers
# ------------------------------------------------------------------------------------------
def q_bf16(x):
    """round-to-nearest-even to bf16, returned as fp32 (what v_cvt_pk_bf16_f32 does)."""
    return x.to(torch.bfloat16).to(torch.float32)


class _Net:
    """Shared pieces of Parser and Tagger."""

    def __init__(self, config, enc, state_dict, ext_word_emb, quant="fp32"):
        assert quant in ("fp32", "bf16")
        if not isinstance(ext_word_emb, (list, tuple)):
            ext_word_emb = [ext_word_emb]
        self.cfg, self.enc, self.quant = config, enc, quant
        self.K = len(ext_word_emb)
        self.sd = {k: v.detach().to(torch.float32) if v.is_floating_point() else v for k, v in state_dict.items()}
        self.q = q_bf16 if quant == "bf16" else (lambda x: x)

    # -- a Linear whose operands are quantised in bf16 mode ---------------------------------
    def lin(self, x, prefix):
        return F.linear(self.q(x), self.q(self.sd[prefix + ".weight"]), self.sd[prefix + ".bias"])

    def emb(self, name, idx):
        return self.q(self.sd[name])[idx]

    def conv_bn(self, x, prefix, in_scale_channels=0):
        """x [S, C, P] -> [S, F, P].  reference: ConvNorm (modules.py:326-344) followed by eval
        BatchNorm1d.  bf16 mode: BN folded into weight/bias, operands rounded.
        `in_scale_channels`: bf16 mode only — the first so many input channels of x were
        stored un-doubled and the factor 2 of WordGram's `x = x + skip` (modules.py:467-468)
        is folded into the weight instead (exact: a power of two)."""
        w, b = self.sd[prefix + ".0.conv.weight"], self.sd[prefix + ".0.conv.bias"]
        g, beta = self.sd[prefix + ".1.weight"], self.sd[prefix + ".1.bias"]
        mu, var = self.sd[prefix + ".1.running_mean"], self.sd[prefix + ".1.running_var"]
        if self.quant == "fp32":
            return F.batch_norm(F.conv1d(x, w, b, padding=2), mu, var, g, beta, False, 0.0, 1e-5)
        s = g / torch.sqrt(var + 1e-5)
        wf = w * s[:, None, None]
        bf = (b - mu) * s + beta
        if in_scale_channels:
            wf = wf.clone()
            wf[:, :in_scale_channels, :] *= 2.0
        return F.conv1d(self.q(x), self.q(wf), bf, padding=2)

    # -- WordGram.forward (modules.py:450-484) ----------------------------------------------
    def wordgram(self, x_char, x_case, x_lang, x_mask, x_word_len):
        xc = self.emb("_word_net._tok_emb.weight", x_char)
        xs = self.emb("_word_net._case_emb.weight", x_case)
        xl = self.emb("_word_net._lang_emb.weight", x_lang)
        x = torch.cat([xc, xs], dim=-1).permute(0, 2, 1)
        xl = xl.unsqueeze(1).repeat(1, x_case.shape[1], 1).permute(0, 2, 1)
        half = self.cfg.char_filter_size // 2
        res, skip = None, None
        for l in range(self.cfg.char_layers):
            pre = "_word_net._convolutions_char.%d" % l
            if self.quant == "fp32":
                if skip is not None:
                    x = x + skip
                co = self.conv_bn(torch.cat([x, xl], dim=1), pre)
            else:
                co = self.conv_bn(torch.cat([x, xl], dim=1), pre, half if skip is not None else 0)
            tmp = torch.tanh(co[:, :half, :]) * torch.sigmoid(co[:, half:, :])
            # bf16 contract: the HIP path has no fp32 residual stream at this level; the masked
            # mean sums the stored (rounded) layer outputs.  fp32 mode: q is the identity.
            res = self.q(tmp) if res is None else res + self.q(tmp)
            skip = tmp
            x = tmp
        x = (self.q(x) + res).permute(0, 2, 1)
        x = x * x_mask.unsqueeze(2)  # float64 mask promotes the product (utils.py:247)
        pre = torch.sum(x, dim=1, dtype=torch.float)
        norm = pre / torch.clip(x_word_len.unsqueeze(1), min=1)
        return torch.tanh(self.lin(norm, "_word_net._pre_out.linear_layer"))

    def ext_proj(self, x_word_emb_packed, sl, L):
        """parser.py:130-138 / tagger.py:111-119; MLP = modules.py:41-52."""
        acc = None
        for i in range(len(x_word_emb_packed)):
            we = unpack(x_word_emb_packed[i], sl, L)
            h = torch.tanh(self.lin(we, "_ext_proj.%d._h1.linear_layer" % i))
            o = self.lin(h, "_ext_proj.%d._h2.linear_layer" % i)
            acc = o if acc is None else acc + o
        return torch.tanh(acc / len(x_word_emb_packed))

    def gated_convs(self, x, lang_emb, aux_location=None):
        """parser.py:157-176 / tagger.py:134-148.  x, lang_emb channel-first."""
        half = self.cfg.cnn_filter // 2
        res, hidden = None, None
        for l in range(self.cfg.cnn_layers):
            co = self.conv_bn(x, "_convs.%d" % l)
            tmp = torch.tanh(co[:, :half, :]) * torch.sigmoid(co[:, half:, :])
            res = tmp if res is None else res + tmp
            x = tmp
            if aux_location is not None and l + 1 == aux_location:
                hidden = torch.cat([x + res, lang_emb], dim=1)
            if l + 1 != self.cfg.cnn_layers:
                x = torch.cat([x, lang_emb], dim=1)
        return x + res, hidden


class OracleParser(_Net):
    """reference: cube/networks/parser.py:22-95 (construction), :109-233, :404-448."""

    # -- nn.LSTM(batch_first, bidirectional) on a padded batch (parser.py:71-72,201) --------
    def lstm(self, x):
        H = self.cfg.rnn_size
        for l in range(self.cfg.rnn_layers):
            outs = []
            for suffix in ("", "_reverse"):
                w_ih = self.sd["_rnn.weight_ih_l%d%s" % (l, suffix)]
                w_hh = self.sd["_rnn.weight_hh_l%d%s" % (l, suffix)]
                bias = self.sd["_rnn.bias_ih_l%d%s" % (l, suffix)] + self.sd["_rnn.bias_hh_l%d%s" % (l, suffix)]
                gx = F.linear(self.q(x), self.q(w_ih), bias)  # [B, T, 4H]
                B, T, _ = gx.shape
                h = torch.zeros((B, H))
                c = torch.zeros((B, H))
                out = torch.zeros((B, T, H))
                steps = range(T - 1, -1, -1) if suffix else range(T)
                whh = self.q(w_hh)
                for t in steps:
                    g = gx[:, t] + F.linear(self.q(h), whh)
                    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
                    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                    h = torch.sigmoid(o) * torch.tanh(c)
                    out[:, t] = h
                outs.append(out)
            x = torch.cat(outs, dim=-1)
        return x

    def forward(self, X, return_intermediates=False):
        cfg = self.cfg
        x_sents = X["x_sent"]
        sl = X["x_sent_len"].numpy()
        L = x_sents.shape[1]
        inter = {}
        char_emb_packed = self.wordgram(X["x_word"], X["x_word_case"], X["x_lang_word"], X["x_word_masks"],
                                        X["x_word_len"])
        inter["char_emb_packed"] = char_emb_packed
        char_emb = unpack(char_emb_packed, sl, L)
        word_emb_ext = self.ext_proj(X["x_word_embeddings"], sl, L)
        inter["word_emb_ext"] = word_emb_ext
        lang_emb = self.emb("_lang_emb.weight", X["x_lang_sent"]).unsqueeze(1).repeat(1, L + 1, 1)
        cl = torch.cat([char_emb, lang_emb[:, 1:, :]], dim=-1)
        aupos = self.lin(cl, "_aupos.linear_layer")
        aattrs = self.lin(cl, "_aattrs.linear_layer")
        axpos = self.lin(cl, "_axpos.linear_layer")
        word_emb = self.emb("_word_emb.weight", x_sents)
        x = torch.cat([word_emb, char_emb, word_emb_ext, lang_emb[:, 1:, :]], dim=-1)
        root = self.q(self.sd["_r_emb.weight"])[torch.zeros((x.shape[0], 1), dtype=torch.long)]
        x = torch.cat([root, x], dim=1).permute(0, 2, 1)
        lang_cf = lang_emb.permute(0, 2, 1)
        x, hidden = self.gated_convs(x, lang_cf, cfg.aux_softmax_location)
        x_parse = x.permute(0, 2, 1)
        inter["x_parse"] = x_parse
        hidden = hidden.permute(0, 2, 1)[:, 1:, :]
        inter["hidden"] = hidden
        pre_morpho = torch.tanh(self.lin(hidden, "_pre_morpho.linear_layer"))
        pre_morpho = torch.cat([pre_morpho, lang_emb[:, 1:, :]], dim=2)
        upos = self.lin(pre_morpho, "_upos.linear_layer")
        upos_idx = X["y_upos"] if "y_upos" in X else torch.argmax(upos, dim=-1)
        upos_emb = self.emb("_upos_emb.weight", upos_idx)
        pm = torch.cat([pre_morpho, upos_emb], dim=-1)
        attrs = self.lin(pm, "_attrs.linear_layer")
        xpos = self.lin(pm, "_xpos.linear_layer")
        ext0 = torch.cat([torch.zeros((x.shape[0], 1, cfg.external_proj_size)), word_emb_ext], dim=1)
        xr = torch.cat([x_parse, ext0, lang_emb], dim=-1)
        output = self.lstm(xr)
        inter["rnn_out"] = output
        output = torch.cat([output, lang_emb], dim=-1)
        pre_parsing = torch.tanh(self.lin(output, "_pre_out.linear_layer"))
        if self.quant == "bf16":
            pre_parsing = self.q(pre_parsing)  # stored as bf16 by the HIP path
        heads = self.arc_scores(pre_parsing)
        out = (heads[:, 1:, :], pre_parsing, upos, xpos, attrs, aupos, axpos, aattrs)
        return (out, inter) if return_intermediates else out

    def arc_scores(self, pre):
        """DeepBiaffine(pairwise) -> PairwiseBiaffineScorer -> PairwiseBilinear
        (modules.py:564-579, 550-561, 518-533): S[b,i,j] = [h1_i;1]^T W [h2_j;1]; the
        PairwiseBilinear bias is never added (modules.py:516 vs 518-533)."""
        h1 = F.relu(self.lin(pre, "_head.W1"))
        h2 = F.relu(self.lin(pre, "_head.W2"))
        one = torch.ones(h1.shape[:-1] + (1,))
        a1 = torch.cat([self.q(h1), one], dim=-1)
        a2 = torch.cat([self.q(h2), one], dim=-1)
        W = self.q(self.sd["_head.scorer.W_bilin.weight"][:, :, 0])
        u = self.q(a1 @ W)  # [B, T, 101]
        return u @ a2.transpose(1, 2)

    def get_labels(self, labels, heads):
        """parser.py:218-233 + BiaffineScorer (modules.py:536-547, nn.Bilinear with bias)."""
        B, T, _ = labels.shape
        idx = torch.zeros((B, T - 1), dtype=torch.long)
        for b in range(B):
            n = len(heads[b])
            idx[b, :n] = torch.as_tensor(np.asarray(heads[b]), dtype=torch.long)
        x2 = torch.gather(labels, 1, idx.unsqueeze(-1).expand(-1, -1, labels.shape[-1]))
        h1 = self.q(F.relu(self.lin(labels[:, 1:, :], "_label.W1")))
        h2 = self.q(F.relu(self.lin(x2, "_label.W2")))
        one = torch.ones(h1.shape[:-1] + (1,))
        a1 = torch.cat([h1, one], dim=-1)
        a2 = torch.cat([h2, one], dim=-1)
        W = self.q(self.sd["_label.scorer.W_bilin.weight"])  # [n_l, 201, 201]
        v = torch.einsum("blp,opq->bloq", a1, W)
        return torch.einsum("bloq,blq->blo", v, a2) + self.sd["_label.scorer.W_bilin.bias"]

    def decode_batch(self, X):
        """One batch of Parser.process (parser.py:418-431): -> dict of numpy int arrays."""
        att, labels, p_upos, p_xpos, p_attrs, _, _, _ = self.forward(X)
        sl = X["x_sent_len"].numpy()
        probs = torch.softmax(att, dim=-1).numpy()
        pred_heads = cle_oracle.decode(probs, sl)
        lab = self.get_labels(labels, pred_heads)
        return {"arc_logits": att.numpy(), "arc_probs": probs, "heads": pred_heads, "label_logits": lab.numpy(),
                "labels": torch.argmax(lab, dim=-1).numpy(), "upos": torch.argmax(p_upos, dim=-1).numpy(),
                "xpos": torch.argmax(p_xpos, dim=-1).numpy(), "attrs": torch.argmax(p_attrs, dim=-1).numpy(),
                "upos_logits": p_upos.numpy(), "xpos_logits": p_xpos.numpy(), "attrs_logits": p_attrs.numpy(),
                "pre_parsing": labels.numpy()}

    def process(self, doc, batch_size=4):
        """Parser.process (parser.py:404-448): in-order batches, results written into `doc`."""
        enc = self.enc
        with torch.no_grad():
            for start in range(0, len(doc.sentences), batch_size):
                sents = doc.sentences[start:start + batch_size]
                r = self.decode_batch(collate(sents, enc))
                for b, s in enumerate(sents):
                    for i, w in enumerate(s.words):
                        w.head = r["heads"][b][i]
                        w.label = enc.labels[r["labels"][b, i]]
                        w.upos = enc.upos_list[r["upos"][b, i]]
                        w.xpos = enc.xpos_list[r["xpos"][b, i]]
                        w.attrs = enc.attrs_list[r["attrs"][b, i]]
        return doc


class OracleTagger(_Net):
    """reference: cube/networks/tagger.py:22-67, 91-162, 288-329."""

    def forward(self, X):
        x_sents = X["x_sent"]
        sl = X["x_sent_len"].numpy()
        L = x_sents.shape[1]
        char_emb = unpack(self.wordgram(X["x_word"], X["x_word_case"], X["x_lang_word"], X["x_word_masks"],
                                        X["x_word_len"]), sl, L)
        word_emb_ext = self.ext_proj(X["x_word_embeddings"], sl, L)
        lang_emb = self.emb("_lang_emb.weight", X["x_lang_sent"]).unsqueeze(1).repeat(1, L, 1)
        cl = torch.cat([char_emb, lang_emb], dim=-1)
        aupos = self.lin(cl, "_aupos.linear_layer")
        axpos = self.lin(cl, "_axpos.linear_layer")
        aattrs = self.lin(cl, "_aattrs.linear_layer")
        word_emb = self.emb("_word_emb.weight", x_sents)
        x = torch.cat([word_emb, char_emb, word_emb_ext, lang_emb], dim=-1).permute(0, 2, 1)
        lang_cf = lang_emb.permute(0, 2, 1)
        x, _ = self.gated_convs(x, lang_cf)
        xu = x.permute(0, 2, 1)
        xl = torch.cat([xu, lang_emb], dim=-1)
        upos = self.lin(xl, "_upos.linear_layer")
        upos_idx = X["y_upos"] if "y_upos" in X else torch.argmax(upos, dim=-1)
        ue = torch.cat([self.emb("_upos_emb.weight", upos_idx), lang_emb, xu], dim=-1)
        xpos = self.lin(ue, "_xpos.linear_layer")
        attrs = self.lin(ue, "_attrs.linear_layer")
        return upos, xpos, attrs, aupos, axpos, aattrs

    def process(self, doc, upos=True, xpos=True, attrs=True, batch_size=32):
        if not (upos or xpos or attrs):
            raise Exception("To perform tagging at least one of 'upos', 'xpos' or 'attrs' must be set to True.")
        enc = self.enc
        with torch.no_grad():
            for start in range(0, len(doc.sentences), batch_size):
                sents = doc.sentences[start:start + batch_size]
                p_u, p_x, p_a, _, _, _ = self.forward(collate(sents, enc))
                u, x, a = (torch.argmax(t, dim=-1).numpy() for t in (p_u, p_x, p_a))
                for b, s in enumerate(sents):
                    for i, w in enumerate(s.words):
                        if upos:
                            w.upos = enc.upos_list[u[b, i]]
                        if xpos:
                            w.xpos = enc.xpos_list[x[b, i]]
                        if attrs:
                            w.attrs = enc.attrs_list[a[b, i]]
        return doc
