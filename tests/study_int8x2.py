"""Design study that preceded cube_gemm_i8x2 (not a test; run by hand: python tests/study_int8x2.py).

Would a two-slice int8 fixed-point split of the gated-convolution GEMMs (int8 MFMA runs at twice the
bf16 rate and moves half the bytes; three slice products instead of three bf16 passes) stay inside the
1e-3 parity bar?  The oracle is run with the operands of every gated Conv1d quantised the way such a
kernel would see them -- activations: 15-bit signed fixed point with one scale per tensor (gate outputs
are bounded by 1), weights: 15-bit signed fixed point with one scale per output channel, exact integer
accumulation -- everything else in fp32, and compared with the fp32 oracle on the full-size model.
(The first character layer is a table look-up in the HIP path and stays exact.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import parser_oracle  # noqa: E402
from tests import cases  # noqa: E402

BITS = int(os.environ.get("BITS", "15"))
QMAX = float(2 ** (BITS - 1) - 1)


def fixed(x, scale):
    return torch.round(x / scale * QMAX).clamp(-QMAX - 1, QMAX) * (scale / QMAX)


class FixedPointConvs(parser_oracle.OracleParser):
    def conv_bn(self, x, prefix, in_scale_channels=0):
        w, b = self.sd[prefix + ".0.conv.weight"], self.sd[prefix + ".0.conv.bias"]
        g, beta = self.sd[prefix + ".1.weight"], self.sd[prefix + ".1.bias"]
        mu, var = self.sd[prefix + ".1.running_mean"], self.sd[prefix + ".1.running_var"]
        s = g / torch.sqrt(var + 1e-5)
        wf = (w * s[:, None, None]).double()
        bf = ((b - mu) * s + beta).double()
        if prefix.endswith("_convolutions_char.0"):  # look-up tables in the HIP path: exact
            return F.conv1d(x.double(), wf, bf, padding=2).float()
        xs = x.abs().max().clamp(min=1e-30)
        ws = wf.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-30)
        return F.conv1d(fixed(x.double(), xs.double()), fixed(wf, ws), bf, padding=2).float()


def main():
    case = cases.FULL
    cfg, enc, sd = cases.parser_model(case)
    for key in ("full_8x32", "g6"):
        try:
            doc = cases.document(case, enc, key)
        except Exception:
            continue
        X = parser_oracle.collate(doc.sentences, enc)
        ref = parser_oracle.OracleParser(cfg, enc, sd, list(case["ext"]), quant="fp32")
        fx = FixedPointConvs(cfg, enc, sd, list(case["ext"]), quant="fp32")
        with torch.no_grad():
            a, b = ref.decode_batch(X), fx.decode_batch(X)
        out = {}
        for k in ("arc_logits", "label_logits", "upos_logits"):
            if k in a:
                out[k] = float(np.abs(a[k] - b[k]).max() / np.abs(a[k]).max())
        heads_same = np.mean([np.mean(np.asarray(x) == np.asarray(y)) for x, y in zip(a["heads"], b["heads"])])
        print("%s  %d-bit fixed point: max rel err %s, heads equal %.4f" % (key, BITS, out, heads_same))


if __name__ == "__main__":
    main()
