"""Design study (CPU, oracle only): what would the FIRST word-level gated convolution cost in decisions if it ran in the
int8x2 fixed point of cube_gemm_i8x2 -- inputs [word | char | lang | ext] as 16-bit fixed point with one scale per block
(embedding-table maximum / 1 for the tanh outputs), weights 16 bits per output channel with the block scales folded in?
The config-1 bench batch against the fp32 oracle.    python scripts/word0_i8_study.py [B L]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import parser_oracle  # noqa: E402
from tests import cases  # noqa: E402
from cube_amd import synth  # noqa: E402

ONE = 32512.0


def fixed(x, scale):
    return torch.round(x / scale * ONE).clamp(-ONE, ONE) * (scale / ONE)


class Word0Fixed(parser_oracle.OracleParser):
    blocks = None

    def conv_bn(self, x, prefix, in_scale_channels=0):
        if prefix != "_convs.0":
            return super().conv_bn(x, prefix, in_scale_channels)
        w, b = self.sd[prefix + ".0.conv.weight"], self.sd[prefix + ".0.conv.bias"]
        g, beta = self.sd[prefix + ".1.weight"], self.sd[prefix + ".1.bias"]
        mu, var = self.sd[prefix + ".1.running_mean"], self.sd[prefix + ".1.running_var"]
        s = g / torch.sqrt(var + 1e-5)
        wf = (w * s[:, None, None]).double()
        bf = ((b - mu) * s + beta).double()
        xq = x.double().clone()
        wq = wf.clone()
        for lo, hi in self.blocks:  # activations: block maximum -> 16 bits; the block scale goes into the weights
            sc = xq[:, lo:hi].abs().max().clamp(min=1e-30)
            xq[:, lo:hi] = fixed(xq[:, lo:hi], sc) / sc
            wq[:, lo:hi] *= sc
        ws = wq.abs().amax(dim=(1, 2), keepdim=True).clamp(min=1e-30)
        return F.conv1d(xq, fixed(wq, ws), bf, padding=2).float()


def main():
    B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 32)
    cfg, enc, sd = cases.parser_model(cases.FULL)
    torch.set_num_threads(8)
    doc = synth.make_document(enc, [L] * B, seed=1000, ext_word_emb=(300,))
    X = parser_oracle.collate(doc.sentences, enc)
    ref = parser_oracle.OracleParser(cfg, enc, sd, [300], quant="fp32")
    fx = Word0Fixed(cfg, enc, sd, [300], quant="fp32")
    cin = sd["_convs.0.0.conv.weight"].shape[1]
    fx.blocks = [(0, 256), (256, 512), (512, 576), (576, cin)]
    with torch.no_grad():
        a, b = ref.decode_batch(X), fx.decode_batch(X)
    d = b["arc_logits"] - a["arc_logits"]
    ha, hb = np.stack([np.asarray(h) for h in a["heads"]]), np.stack([np.asarray(h) for h in b["heads"]])
    print("word layer 0 in 16-bit fixed point: arc rms %.2e max %.2e (relative to the largest score); heads differ: %d of %d"
          % (np.sqrt((d ** 2).mean()) / np.abs(a["arc_logits"]).max(), np.abs(d).max() / np.abs(a["arc_logits"]).max(),
             int((ha != hb).sum()), ha.size))


if __name__ == "__main__":
    main()
