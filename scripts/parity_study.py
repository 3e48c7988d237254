"""The two full-size bench batches (config 1: 256 x 32, config 2: 64 x 128; 8192 tokens each) against the REFERENCE's
own outputs (tests/golden/g11_bench_*.npz), with the fixed-point switches on and off: score errors and how many
heads / labels / UPOS tags differ.  north_star: >= 99.9 % of the tokens, i.e. at most 8 of 8192.

    python scripts/parity_study.py [config ...]         (GPU)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from tests import cases  # noqa: E402
from cube_amd import synth  # noqa: E402
from cube_amd.networks.parser import Parser  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402

dev = torch.device("cuda:0")
cfg, enc, sd = cases.parser_model(cases.FULL)
col = MorphoCollate(enc)
MODES = (("all fixed point", {}),
         ("word bf16x3", dict(word_int8=False)),
         ("lstm bf16x3", dict(lstm_int8=False)),
         ("char bf16x3", dict(char_int8=False)),
         ("word+lstm bf16x3", dict(word_int8=False, lstm_int8=False)),
         ("all bf16x3", dict(word_int8=False, lstm_int8=False, char_int8=False)))


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


for tag, B, L in (("config1_256x32", 256, 32), ("config2_64x128", 64, 128)):
    if len(sys.argv) > 1 and tag[:7] not in sys.argv[1:]:
        continue
    g = np.load(os.path.join(ROOT, "tests", "golden", "g11_bench_%s.npz" % tag))
    doc = synth.make_document(enc, [L] * B, seed=1000, ext_word_emb=(300,))
    X = col.collate_fn(doc.sentences, targets=False)
    print("==== %s" % tag, flush=True)
    for name, flags in MODES:
        p = Parser(cfg, enc, ext_word_emb=[300])
        for k, v in flags.items():
            setattr(p, k, v)
        p.load_state_dict(sd)
        p.to(dev)
        r = p.decode_batch(X)
        heads = r["heads"].cpu().numpy()
        dh = heads != g["heads"]
        same = ~dh
        dl = (r["labels"].cpu().numpy() != g["labels"]) & same
        du = r["upos"].cpu().numpy() != g["upos"]
        arc = r["arc_logits"].cpu().numpy()
        d = arc - g["arc_logits"]
        print("%-18s arc max %.2e rms %.2e  upos %.2e | heads differ %2d tokens / %2d sentences, labels (same head) %d, upos %d"
              % (name, rel(arc, g["arc_logits"]), np.sqrt((d ** 2).mean()) / np.abs(g["arc_logits"]).max(),
                 rel(r["upos_logits"].cpu().numpy(), g["upos_logits"]), int(dh.sum()), int(dh.any(axis=1).sum()),
                 int(dl.sum()), int(du.sum())), flush=True)
        del p
