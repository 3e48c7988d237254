"""ORACLE (test infrastructure, never shipped, never measured as the product).

CPU restatement (PyTorch-CPU fp32) of the reference tokenizer's network and of the label decoding
of `process`:

  Tokenizer.forward           cube/networks/tokenizer.py:70-127  (WordGram: modules.py:450-484,
                                                                  unpack: utils.py:17-32)
  Tokenizer.process (decode)  cube/networks/tokenizer.py:251-304

The sub-word collate (`TokenCollateHF`, utils_tokenizer.py:361-576) needs a HuggingFace tokenizer and
LM that are not available offline; batches come from `cube_amd.synth.make_token_batch` (same dict
format).  Pinned against the reference's `forward` by tests/golden/g10_tokenizer.npz.
Only tests/ may import this.
"""
import torch
import torch.nn.functional as F

from oracle import parser_oracle


class OracleTokenizer(parser_oracle._Net):
    def __init__(self, config, enc, state_dict, ext_word_emb):
        super().__init__(config, enc, state_dict, ext_word_emb, quant="fp32")

    def wordgram(self, x_char, x_case, x_lang, x_mask, x_word_len):
        """WordGram with 3 layers under the `_wg` prefix (tokenizer.py:49,81)."""
        sd = self.sd
        xc, xs, xl = sd["_wg._tok_emb.weight"][x_char], sd["_wg._case_emb.weight"][x_case], sd["_wg._lang_emb.weight"][x_lang]
        x = torch.cat([xc, xs], dim=-1).permute(0, 2, 1)
        xl = xl.unsqueeze(1).repeat(1, x_case.shape[1], 1).permute(0, 2, 1)
        res = skip = None
        for l in range(3):
            if skip is not None:
                x = x + skip
            co = self.conv_bn(torch.cat([x, xl], dim=1), "_wg._convolutions_char.%d" % l)
            tmp = torch.tanh(co[:, :256, :]) * torch.sigmoid(co[:, 256:, :])
            res = tmp if res is None else res + tmp
            skip = tmp
            x = tmp
        x = (x + res).permute(0, 2, 1) * x_mask.unsqueeze(2)
        pre = torch.sum(x, dim=1, dtype=torch.float)
        norm = pre / torch.clip(x_word_len.unsqueeze(1), min=1)
        return torch.tanh(F.linear(norm, sd["_wg._pre_out.linear_layer.weight"], sd["_wg._pre_out.linear_layer.bias"]))

    def forward(self, X):
        """-> logits [B, T, 5] (tokenizer.py:70-127)."""
        sd, cfg = self.sd, self.cfg
        with torch.no_grad():
            x_emb = X["x_input"]
            T = x_emb[0].shape[1]
            lang = sd["_lang_emb.weight"][X["x_lang"]].unsqueeze(1).repeat(1, T, 1)
            ce = self.wordgram(X["x_word_char"], X["x_word_case"], X["x_word_lang"], X["x_word_masks"], X["x_word_len"])
            x_char = parser_oracle.unpack(ce, X["x_sent_len"].numpy(), T)
            acc = None
            for i, we in enumerate(x_emb):
                h = torch.tanh(self.lin(we.float(), "_ext_proj.%d._h1.linear_layer" % i))
                o = self.lin(h, "_ext_proj.%d._h2.linear_layer" % i)
                acc = o if acc is None else acc + o
            ext = torch.tanh(acc / len(x_emb))
            x = torch.cat([ext, x_char, sd["_spa_emb.weight"][X["x_input_spa"]], lang], dim=-1).permute(0, 2, 1)
            xl = lang.permute(0, 2, 1)
            half = cfg.cnn_filter // 2
            res = skip = None
            for l in range(cfg.cnn_layers):
                co = self.conv_bn(x, "_convs.%d" % l)
                tmp = torch.tanh(co[:, :half, :]) * torch.sigmoid(co[:, half:, :])
                res = tmp if res is None else res + tmp
                x = tmp
                if l + 1 != cfg.cnn_layers:
                    if skip is not None:
                        x = x + skip
                    skip = x
                    x = torch.cat([x, xl], dim=1)
            x = torch.cat([x + res, xl], dim=1).permute(0, 2, 1)
            return self.lin(x, "_output.linear_layer")


def decode_labels(toks, preds):
    """tokenizer.py:258-287: (sub-word pieces, labels 0..4) -> (sentences of words, multi-word flags).
    2 = end of token, 3 = end of a multi-word token, 4 = end of token and of sentence."""
    p_sents, p_mwes, p_sent, p_mwe, tok = [], [], [], [], ""
    for pred, text in zip(preds, toks):
        tok += text.replace("▁", "")
        if pred in (2, 3, 4):
            if tok.strip() != "":
                p_sent.append(tok)
                p_mwe.append(pred == 3)
            tok = ""
        if pred == 4:
            if p_sent:
                p_sents.append(p_sent)
                p_mwes.append(p_mwe)
            p_sent, p_mwe = [], []
    if tok.strip() != "":
        p_sent.append(tok)
        p_mwe.append(False)
    if p_sent:
        p_sents.append(p_sent)
        p_mwes.append(p_mwe)
    return p_sents, p_mwes
