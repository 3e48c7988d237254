"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement, in PyTorch-CPU fp32 + NumPy, of the reference's lemmatizer inference path
(SURVEY.md §8f row 2):

  LemmaDataset (inference)      cube/networks/utils.py:99-122
  Word2TargetCollate.collate_fn cube/networks/utils.py:153-223
  Attention.forward             cube/networks/modules.py:133-169
  Lemmatizer.forward            cube/networks/lemmatizer.py:78-171
  Lemmatizer.process            cube/networks/lemmatizer.py:185-231

Pinned against the reference itself by tests/golden/g9_lemmatizer.npz / g9_lemmas_*.txt
(generated in the build container by tests/golden/make_golden.py g9 with the reference
imported).  Only tests/, __graft_entry__.smoke() and bench legs that time a CPU baseline may
import this.
"""
import numpy as np
import torch
import torch.nn.functional as F


def dataset(doc):
    """LemmaDataset(doc, for_training=False): one example per word, words longer than 25
    characters are skipped (utils.py:109-110)."""
    out = []
    for sent in doc.sentences:
        for w in sent.words:
            if len(w.word) > 25:
                continue
            out.append({"word": w.word, "upos": w.upos, "lang_id": sent.lang_id, "target": w.lemma})
    return out


def _case(ch):
    if ch.lower() == ch.upper():
        return 1
    return 2 if ch.lower() != ch else 3


def collate(batch, enc):
    """Word2TargetCollate.collate_fn without the y_* targets (`process` deletes them,
    lemmatizer.py:200-201)."""
    C = len(enc.char2int)
    n = len(batch)
    S = max(len(e["word"]) for e in batch) + 2
    x_char = np.zeros((n, S), dtype=np.int64)
    x_case = np.zeros((n, S), dtype=np.int64)
    x_lang = np.zeros(n, dtype=np.int64)
    x_upos = np.zeros(n, dtype=np.int64)
    for i, e in enumerate(batch):
        x_lang[i] = e["lang_id"] + 1
        x_upos[i] = enc.upos2int.get(e.get("upos", ""), enc.upos2int["<UNK>"])
        x_char[i, 0] = C  # start index
        p = 1
        for ch in e["word"]:
            x_case[i, p] = _case(ch)
            x_char[i, p] = enc.char2int.get(ch.lower(), enc.char2int["<UNK>"])
            p += 1
        x_char[i, p] = C + 1  # stop index
    return {"x_char": torch.from_numpy(x_char), "x_case": torch.from_numpy(x_case),
            "x_lang": torch.from_numpy(x_lang), "x_upos": torch.from_numpy(x_upos)}


class OracleLemmatizer:
    def __init__(self, config, enc, state_dict, dtype=torch.float32):
        """dtype=torch.float64 runs the same arithmetic in double precision (used by the tests to tell
        a genuine mismatch from a near-tie of the greedy arg-max)."""
        self.cfg, self.enc, self.dt = config, enc, dtype
        self.sd = {k: v.detach().to(dtype) if v.is_floating_point() else v for k, v in state_dict.items()}
        self.eol = len(enc.char2int)

    def _lstm_step(self, x, state):
        """one time step of nn.LSTM(batch_first, unidirectional), all layers (gate order i,f,g,o)."""
        h_out, c_out = [], []
        for l in range(self.cfg.decoder_layers):
            h, c = state[0][l], state[1][l]
            g = (F.linear(x, self.sd["_decoder.weight_ih_l%d" % l], self.sd["_decoder.bias_ih_l%d" % l])
                 + F.linear(h, self.sd["_decoder.weight_hh_l%d" % l], self.sd["_decoder.bias_hh_l%d" % l]))
            i, f, gg, o = g.chunk(4, dim=-1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            h_out.append(h)
            c_out.append(c)
            x = h
        return x, (h_out, c_out)

    def encode(self, X):
        """lemmatizer.py:89-124 -> encoder_output [n, S, 256 + 16 + upos + lang]."""
        sd = self.sd
        S = X["x_char"].shape[1]
        char_emb = sd["_char_emb.weight"][X["x_char"]]
        case_emb = sd["_case_emb.weight"][X["x_case"]]
        upos_emb = sd["_upos_emb.weight"][X["x_upos"]].unsqueeze(1).repeat(1, S, 1)
        lang_emb = sd["_lang_emb.weight"][X["x_lang"]].unsqueeze(1).repeat(1, S, 1)
        cond = torch.cat([case_emb, upos_emb], dim=-1)
        x = torch.cat([char_emb, cond], dim=-1).permute(0, 2, 1)
        xl = lang_emb.permute(0, 2, 1)
        half = 256
        res = skip = None
        for l in range(5):
            if skip is not None:
                x = x + skip
            p = "_convolutions_char.%d" % l
            co = F.conv1d(torch.cat([x, xl], dim=1), sd[p + ".0.conv.weight"], sd[p + ".0.conv.bias"], padding=2)
            co = F.batch_norm(co, sd[p + ".1.running_mean"], sd[p + ".1.running_var"], sd[p + ".1.weight"],
                              sd[p + ".1.bias"], False, 0.0, 1e-5)
            tmp = torch.tanh(co[:, :half, :]) * torch.sigmoid(co[:, half:, :])
            res = tmp if res is None else res + tmp
            skip = tmp
            x = tmp
        x = (x + res).permute(0, 2, 1)
        return torch.cat([x, cond, lang_emb], dim=-1)

    def attention(self, hidden, enc_out):
        """modules.py:145-169 (soft-max over ALL source positions, pads included)."""
        S = enc_out.shape[1]
        h = hidden.unsqueeze(1).repeat(1, S, 1)
        e = F.conv1d(torch.cat([h, enc_out], dim=2).transpose(1, 2), self.sd["_attention.attn.conv.weight"],
                     self.sd["_attention.attn.conv.bias"], padding=2)
        e = torch.tanh(e)                                # [n, proj, S]
        a = torch.einsum("p,nps->ns", self.sd["_attention.v"], e)
        return F.softmax(a, dim=1)

    def forward(self, X, max_steps=None, return_margins=False):
        """-> (char logits [n, steps, C+2], case logits [n, steps, 4]) exactly as the reference's
        inference branch: greedy, stops once every word has emitted EOL or after 20*S steps."""
        sd = self.sd
        with torch.no_grad():
            enc_out = self.encode(X)
            n, S = enc_out.shape[0], enc_out.shape[1]
            L = self.cfg.decoder_layers
            D = self.cfg.decoder_size
            zero = ([torch.zeros(n, D, dtype=self.dt) for _ in range(L)], [torch.zeros(n, D, dtype=self.dt) for _ in range(L)])
            _, state = self._lstm_step(sd["_start_frame.weight"][0].unsqueeze(0).repeat(n, 1), zero)
            done = np.zeros(n, dtype=bool)
            prev = torch.zeros(n, self.cfg.char_emb_size, dtype=self.dt)
            oc, ok = [], []
            step = 0
            limit = S * 20 if max_steps is None else max_steps
            while not done.all() and step < limit:
                # the query is `decoder_hidden[-1][-1]` (lemmatizer.py:144): decoder_hidden = (h_n, c_n),
                # so this is the CELL state of the top layer, not its hidden state
                att = self.attention(state[1][-1], enc_out)
                ctx = torch.bmm(att.unsqueeze(1), enc_out).squeeze(1)
                out, state = self._lstm_step(torch.cat([ctx, prev], dim=-1), state)
                lc = F.linear(out, sd["_output_char.linear_layer.weight"], sd["_output_char.linear_layer.bias"])
                lk = F.linear(out, sd["_output_case.linear_layer.weight"], sd["_output_case.linear_layer.bias"])
                oc.append(lc)
                ok.append(lk)
                sel = torch.argmax(lc, dim=-1)
                done |= (sel == self.eol).numpy()
                prev = sd["_char_emb.weight"][sel]
                step += 1
            return torch.stack(oc, dim=1), torch.stack(ok, dim=1)

    def lemmas(self, y_char, y_case):
        """lemmatizer.py:208-224: arg-max characters up to the first EOL, case head applied."""
        out = []
        yc = torch.argmax(y_char, dim=-1).numpy()
        yk = torch.argmax(y_case, dim=-1).numpy()
        for i in range(yc.shape[0]):
            s = []
            for c, k in zip(yc[i], yk[i]):
                if c == self.eol:
                    break
                ch = self.enc.characters[c]
                if k == 2:
                    ch = ch.upper()
                elif k == 3:
                    ch = ch.lower()
                s.append(ch)
            out.append("".join(s))
        return out

    def process(self, doc, batch_size=4):
        """lemmatizer.py:185-231 (lemmas are assigned to the words in order, :225-229)."""
        ex = dataset(doc)
        all_lemmas = []
        for s in range(0, len(ex), batch_size):
            X = collate(ex[s:s + batch_size], self.enc)
            all_lemmas.extend(self.lemmas(*self.forward(X)))
        i = 0
        for sent in doc.sentences:
            for w in sent.words:
                w.lemma = all_lemmas[i]
                i += 1
        return doc
