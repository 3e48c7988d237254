"""Design study that preceded the int8x2 recurrent kernel (not a test; run by hand: python tests/study_lstm_int8x2.py).

Can the recurrent product h_{t-1} . W_hh^T of the BiLSTM run in two-slice int8 fixed point (the scheme of
cube_gemm_i8x2: x = (128 hi + lo) / 16256, |h| < 1 because h = o * tanh(c); weights with one scale per gate row;
hi*hi and hi*lo + lo*hi accumulated exactly, lo*lo dropped) and stay inside the 1e-3 parity bar?  Only that
product is quantised; the input projections, gates and cell state stay fp32, as in the kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import parser_oracle  # noqa: E402
from tests import cases  # noqa: E402

ONE = 16256.0
WBITS_LO = int(os.environ.get("WLO", "7"))   # bits of the weights' low slice: 7 = int8x2 (15-bit), 4 = 12-bit weights
WSTEP = float(2 ** WBITS_LO)
WONE = 127.0 * WSTEP


def slices(x, scale):
    """x / scale in [-1, 1] -> (hi, lo) integers with q = 128 hi + lo = round(x / scale * ONE), lo in [-64, 63]."""
    q = torch.round(x / scale * ONE).clamp(-ONE, ONE)
    hi = torch.floor((q + 64) / 128)
    return hi, q - 128 * hi


class Int8x2Recurrence(parser_oracle.OracleParser):
    def lstm(self, x):
        H = self.cfg.rnn_size
        for l in range(self.cfg.rnn_layers):
            outs = []
            for suffix in ("", "_reverse"):
                w_ih = self.sd["_rnn.weight_ih_l%d%s" % (l, suffix)]
                w_hh = self.sd["_rnn.weight_hh_l%d%s" % (l, suffix)].double()
                bias = self.sd["_rnn.bias_ih_l%d%s" % (l, suffix)] + self.sd["_rnn.bias_hh_l%d%s" % (l, suffix)]
                gx = F.linear(x, w_ih, bias)
                B, T, _ = gx.shape
                h = torch.zeros((B, H))
                c = torch.zeros((B, H))
                out = torch.zeros((B, T, H))
                wmax = w_hh.abs().amax(dim=1, keepdim=True).clamp(min=1e-30)
                qw = torch.round(w_hh / wmax * WONE).clamp(-WONE, WONE)
                whi = torch.floor((qw + WSTEP / 2) / WSTEP)
                wlo = qw - WSTEP * whi
                steps = range(T - 1, -1, -1) if suffix else range(T)
                for t in steps:
                    hhi, hlo = slices(h.double(), torch.tensor(1.0, dtype=torch.float64))
                    acc = 128.0 * WSTEP * (hhi @ whi.T) + 128.0 * (hhi @ wlo.T) + WSTEP * (hlo @ whi.T)   # exact integers, lo*lo dropped
                    rec = (acc * (wmax.T / (ONE * WONE))).float()
                    g = gx[:, t] + rec
                    i, f, gg, o = g[:, :H], g[:, H:2 * H], g[:, 2 * H:3 * H], g[:, 3 * H:]
                    c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
                    h = torch.sigmoid(o) * torch.tanh(c)
                    out[:, t] = h
                outs.append(out)
            x = torch.cat(outs, dim=-1)
        return x


def main():
    case = cases.FULL
    cfg, enc, sd = cases.parser_model(case)
    torch.set_num_threads(8)
    for key in ("full_8x32", "ragged", "full_2x128"):
        doc = cases.document(case, enc, key)
        X = parser_oracle.collate(doc.sentences, enc)
        ref = parser_oracle.OracleParser(cfg, enc, sd, list(case["ext"]), quant="fp32")
        fx = Int8x2Recurrence(cfg, enc, sd, list(case["ext"]), quant="fp32")
        with torch.no_grad():
            a, b = ref.decode_batch(X), fx.decode_batch(X)
        rel = {k: float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max()) for k in ("arc_logits", "label_logits", "pre_parsing")}
        heads = np.mean([np.array_equal(x, y) for x, y in zip(a["heads"], b["heads"])])
        print(key, {k: "%.2e" % v for k, v in rel.items()}, "sentences with equal heads: %.3f" % heads, flush=True)


if __name__ == "__main__":
    main()
