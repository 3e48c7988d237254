"""Design study (CPU, oracle only): which stored tensor of the back stage (BiLSTM -> pre_parsing -> arc scorer) costs how
much arc-score precision when it is kept as a bf16 hi/lo pair (16 significant bits, the bf16x3 kernels) or in the
fixed-point formats of csrc/lstm_i8.hip?  The LSTM input comes from ONE fp32 oracle pass; "truth" = float64 back stage.

    python scripts/back_prec_study.py [B L]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import parser_oracle, cle_oracle  # noqa: E402
from tests import cases  # noqa: E402
from cube_amd import synth  # noqa: E402


def pair(x):
    """x -> bf16 hi + bf16 lo (what a hi/lo plane pair holds), same dtype as x"""
    hi = x.to(torch.bfloat16).to(x.dtype)
    return hi + (x - hi).to(torch.bfloat16).to(x.dtype)


def ident(x):
    return x


def fixed(shift):
    one = 127.0 * 2 ** shift

    def f(x):
        return torch.round(x * one).clamp(-one, one) / one
    return f


class Capture(parser_oracle.OracleParser):
    def lstm(self, x):
        self.lstm_in = x.clone()
        return super().lstm(x)


def back(net, x, lang_emb, q, dt=torch.float64):
    """q: dict of rounding functions for x_in, w_ih, w_hh, h (state fed back), h_out, w (dense weights), pp, hal, u."""
    g = lambda k: q.get(k, ident)  # noqa: E731
    H = net.cfg.rnn_size
    x = g("x_in")(x.to(dt))
    for l in range(net.cfg.rnn_layers):
        outs = []
        for suffix in ("", "_reverse"):
            w_ih = g("w_ih")(net.sd["_rnn.weight_ih_l%d%s" % (l, suffix)].to(dt))
            w_hh = g("w_hh")(net.sd["_rnn.weight_hh_l%d%s" % (l, suffix)].to(dt))
            bias = (net.sd["_rnn.bias_ih_l%d%s" % (l, suffix)] + net.sd["_rnn.bias_hh_l%d%s" % (l, suffix)]).to(dt)
            gx = F.linear(x, w_ih, bias)
            gx = g("gx")(gx)
            B, T, _ = gx.shape
            h = torch.zeros((B, H), dtype=dt)
            c = torch.zeros((B, H), dtype=dt)
            out = torch.zeros((B, T, H), dtype=dt)
            for t in (range(T - 1, -1, -1) if suffix else range(T)):
                gg_ = gx[:, t] + F.linear(g("h")(h), w_hh)
                i, f, gg, o = gg_[:, :H], gg_[:, H:2 * H], gg_[:, 2 * H:3 * H], gg_[:, 3 * H:]
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                out[:, t] = h
            outs.append(out)
        x = g("h_out")(torch.cat(outs, dim=-1))
    W = lambda k: g("w")(net.sd[k].to(dt))  # noqa: E731
    xo = torch.cat([x, g("x_in")(lang_emb.to(dt))], dim=-1)
    pp = g("pp")(torch.tanh(F.linear(xo, W("_pre_out.linear_layer.weight"), net.sd["_pre_out.linear_layer.bias"].to(dt))))
    h1 = g("hal")(F.relu(F.linear(pp, W("_head.W1.weight"), net.sd["_head.W1.bias"].to(dt))))
    h2 = g("hal")(F.relu(F.linear(pp, W("_head.W2.weight"), net.sd["_head.W2.bias"].to(dt))))
    one = torch.ones(h1.shape[:-1] + (1,), dtype=dt)
    a1, a2 = torch.cat([h1, one], dim=-1), torch.cat([h2, one], dim=-1)
    u = g("u")(a1 @ W("_head.scorer.W_bilin.weight")[:, :, 0])
    return (u @ a2.transpose(1, 2))[:, 1:, :]


def main():
    B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 32)
    cfg, enc, sd = cases.parser_model(cases.FULL)
    torch.set_num_threads(8)
    doc = synth.make_document(enc, [L] * B, seed=1000, ext_word_emb=(300,))
    X = parser_oracle.collate(doc.sentences, enc)
    net = Capture(cfg, enc, sd, [300], quant="fp32")
    with torch.no_grad():
        net.forward(X)
        lang_emb = net.emb("_lang_emb.weight", X["x_lang_sent"]).unsqueeze(1).repeat(1, L + 1, 1)
        sl = X["x_sent_len"].numpy()

        def run(q, dt=torch.float64):
            att = back(net, net.lstm_in, lang_emb, q, dt).double()
            heads = cle_oracle.decode(torch.softmax(att.float(), dim=-1).numpy(), sl)
            return att.numpy(), np.stack([np.asarray(h) for h in heads])

        truth, th = run({})
        scale = np.abs(truth).max()
        allp = {k: pair for k in ("x_in", "w_ih", "w_hh", "h", "h_out", "w", "pp", "hal", "u")}
        variants = [("fp32 everywhere", {}, torch.float32)]
        for k in ("x_in", "w_ih", "w_hh", "h", "h_out", "w", "pp", "hal", "u"):
            variants.append(("pair: %s only" % k, {k: pair}, torch.float64))
        variants += [("pair: everything (bf16x3 kernels)", allp, torch.float64),
                     ("pair everything, u exact", {**allp, "u": ident}, torch.float64),
                     ("pair everything, pp/hal/u exact", {**allp, "u": ident, "pp": ident, "hal": ident}, torch.float64),
                     ("pair everything, h = 15-bit fixed", {**allp, "h": fixed(7)}, torch.float64),
                     ("pair everything, h = 16-bit fixed", {**allp, "h": fixed(8)}, torch.float64),
                     ("pair everything, h = 24-bit fixed", {**allp, "h": fixed(16)}, torch.float64)]
        for name, q, dt in variants:
            a, h = run(q, dt)
            d = a - truth
            print("%-40s arc rms %.2e  max %.2e   heads differ: %d tokens in %d sentences" % (
                name, np.sqrt((d ** 2).mean()) / scale, np.abs(d).max() / scale, int((h != th).sum()),
                int((h != th).any(axis=1).sum())), flush=True)


if __name__ == "__main__":
    main()
