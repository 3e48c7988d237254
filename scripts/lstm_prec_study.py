"""Design study (CPU, oracle only; not part of the product): how many bits do h and W_hh need in the fixed-point
recurrence of csrc/lstm_i8.hip before its decisions are as close to the reference as the bf16x3 kernels'?

The LSTM input of one bench batch is taken from ONE fp32 oracle pass; every variant then re-runs only the
recurrence, pre_parsing, the arc scorer and the decoder.  "truth" = the recurrence in float64.  Reported per
variant: rms / max arc-score error relative to the largest score, heads that differ from truth's.

    python scripts/lstm_prec_study.py [B L]
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


class Capture(parser_oracle.OracleParser):
    def lstm(self, x):
        self.lstm_in = x.clone()
        return super().lstm(x)


def recurrence(net, x, mode):
    """mode: dict(dtype=float64|float32, h_shift, w_lo_bits, ll) or None for plain floating point."""
    H = net.cfg.rnn_size
    dt = mode.get("dtype", torch.float32)
    for l in range(net.cfg.rnn_layers):
        outs = []
        for suffix in ("", "_reverse"):
            w_ih = net.sd["_rnn.weight_ih_l%d%s" % (l, suffix)]
            w_hh = net.sd["_rnn.weight_hh_l%d%s" % (l, suffix)]
            bias = net.sd["_rnn.bias_ih_l%d%s" % (l, suffix)] + net.sd["_rnn.bias_hh_l%d%s" % (l, suffix)]
            gx = F.linear(x.to(dt), w_ih.to(dt), bias.to(dt))
            B, T, _ = gx.shape
            h = torch.zeros((B, H), dtype=dt)
            c = torch.zeros((B, H), dtype=dt)
            out = torch.zeros((B, T, H), dtype=dt)
            fixed = "w_lo_bits" in mode
            if fixed:
                hs, wb = mode["h_shift"], mode["w_lo_bits"]
                hone, wstep = 127.0 * 2 ** hs, float(2 ** wb)
                wone = 127.0 * wstep
                w64 = w_hh.double()
                wmax = w64.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
                qw = torch.round(w64 / wmax * wone).clamp(-wone, wone)
                whi = torch.floor((qw + wstep / 2) / wstep)
                wlo = qw - wstep * whi
            steps = range(T - 1, -1, -1) if suffix else range(T)
            for t in steps:
                if fixed and mode.get("h3"):
                    # three 8-bit slices of h (24 bits): a*65536 + b*256 + c; terms = which slice products are kept
                    hone3 = 127.0 * 65536
                    q = torch.round(h.double() * hone3).clamp(-hone3, hone3)
                    a = torch.floor((q + 32768) / 65536)
                    r1 = q - 65536 * a
                    b = torch.floor((r1 + 128) / 256)
                    c_ = r1 - 256 * b
                    hs_ = ((a, 65536.0), (b, 256.0), (c_, 1.0))
                    ws_ = ((whi, wstep), (wlo, 1.0))
                    acc = 0.0
                    for (i, j) in mode["terms"]:
                        acc = acc + hs_[i][1] * ws_[j][1] * (hs_[i][0] @ ws_[j][0].T)
                    rec = (acc * (wmax.T / (hone3 * wone))).float()
                elif fixed:
                    q = torch.round(h.double() * hone).clamp(-hone, hone)
                    hhi = torch.floor((q + 2 ** (hs - 1)) / 2 ** hs)
                    hlo = q - 2 ** hs * hhi
                    acc = 2 ** hs * wstep * (hhi @ whi.T) + 2 ** hs * (hhi @ wlo.T) + wstep * (hlo @ whi.T)
                    if mode.get("ll"):
                        acc = acc + hlo @ wlo.T
                    rec = (acc * (wmax.T / (hone * wone))).float()
                else:
                    rec = F.linear(h, w_hh.to(dt))
                g = gx[:, t] + rec
                i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
                c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                h = torch.sigmoid(o) * torch.tanh(c)
                out[:, t] = h
            outs.append(out)
        x = torch.cat(outs, dim=-1)
    return x.float()


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

        def back(mode):
            out = recurrence(net, net.lstm_in, mode)
            pre = torch.tanh(net.lin(torch.cat([out, lang_emb], dim=-1), "_pre_out.linear_layer"))
            att = net.arc_scores(pre)[:, 1:, :]
            heads = cle_oracle.decode(torch.softmax(att, dim=-1).numpy(), sl)
            return att.numpy(), np.stack([np.asarray(h) for h in heads])

        truth, th = back(dict(dtype=torch.float64))
        scale = np.abs(truth).max()
        variants = [("fp32 floating point", dict()),
                    ("h 7+7  w 8+4 (round 2 kernel)", dict(h_shift=7, w_lo_bits=4)),
                    ("h 8+8  w 8+4", dict(h_shift=8, w_lo_bits=4)),
                    ("h 8+8  w 8+6", dict(h_shift=8, w_lo_bits=6)),
                    ("h 8+8  w 8+6 +LL", dict(h_shift=8, w_lo_bits=6, ll=True)),
                    ("h 8+8  w 8+8", dict(h_shift=8, w_lo_bits=8)),
                    ("h 7+7  w 8+6", dict(h_shift=7, w_lo_bits=6)),
                    ("h 8+8+8 w 8+4: aH aL bH cH", dict(h_shift=8, w_lo_bits=4, h3=True, terms=((0, 0), (0, 1), (1, 0), (2, 0)))),
                    ("h 8+8+8 w 8+6: aH aL bH cH", dict(h_shift=8, w_lo_bits=6, h3=True, terms=((0, 0), (0, 1), (1, 0), (2, 0)))),
                    ("h 8+8+8 w 8+6: aH aL bH cH bL", dict(h_shift=8, w_lo_bits=6, h3=True, terms=((0, 0), (0, 1), (1, 0), (2, 0), (1, 1)))),
                    ("h 8+8+8 w 8+8: aH aL bH cH bL", dict(h_shift=8, w_lo_bits=8, h3=True, terms=((0, 0), (0, 1), (1, 0), (2, 0), (1, 1)))),
                    ("h 8+8+8 w 8+6: aH aL bH bL", dict(h_shift=8, w_lo_bits=6, h3=True, terms=((0, 0), (0, 1), (1, 0), (1, 1))))]
        for name, mode in variants:
            a, h = back(mode)
            d = a - truth
            print("%-32s arc rms %.2e  max %.2e   heads differ: %d tokens in %d sentences" % (
                name, np.sqrt((d ** 2).mean()) / scale, np.abs(d).max() / scale, int((h != th).sum()),
                int((h != th).any(axis=1).sum())), flush=True)


if __name__ == "__main__":
    main()
