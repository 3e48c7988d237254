"""Config-2 batch (64 x 128) against the fp32 oracle with the int8x2 switches on/off: score errors and head agreement."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
from oracle import parser_oracle  # noqa: E402
from tests import cases  # noqa: E402
from cube_amd import synth  # noqa: E402
from cube_amd.networks.parser import Parser  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402

dev = torch.device("cuda:0")
cfg, enc, sd = cases.parser_model(cases.FULL)
B, L = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
doc = synth.make_document(enc, [L] * B, seed=1000, ext_word_emb=(300,))
col = MorphoCollate(enc)
X = col.collate_fn(doc.sentences, targets=False)
ora = parser_oracle.OracleParser(cfg, enc, sd, [300], quant="fp32")
torch.set_num_threads(16)
with torch.no_grad():
    e = ora.decode_batch(parser_oracle.collate(doc.sentences, enc))
ref_heads = np.stack([np.asarray(h) for h in e["heads"]])


def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())


def rel_elem(a, b, floor):
    """element-wise relative error with a floor on the denominator (a fraction of the largest score)"""
    return float((np.abs(a - b) / np.maximum(np.abs(b), floor * np.abs(b).max())).max())


for name, flags in (("all int8x2", {}), ("word bf16x3", dict(word_int8=False)), ("word+lstm bf16x3", dict(word_int8=False, lstm_int8=False)),
                    ("all bf16x3", dict(word_int8=False, lstm_int8=False, char_int8=False))):
    p = Parser(cfg, enc, ext_word_emb=[300])
    for k, v in flags.items():
        setattr(p, k, v)
    p.load_state_dict(sd)
    p.to(dev)
    r = p.decode_batch(X)
    heads = r["heads"].cpu().numpy()
    print("%-18s arc %.2e  label %.2e  upos %.2e  heads equal %.5f (%d tokens in %d sentences differ)" % (
        name, rel(r["arc_logits"].cpu().numpy(), e["arc_logits"]), rel(r["label_logits"].cpu().numpy(), e["label_logits"]),
        rel(r["upos_logits"].cpu().numpy(), e["upos_logits"]), (heads == ref_heads).mean(), int((heads != ref_heads).sum()),
        int((heads != ref_heads).any(axis=1).sum())), flush=True)
    a = r["arc_logits"].cpu().numpy()
    print("%-18s arc element-wise, denominator floored at 10 %% / 1 %% of the largest score: %.2e / %.2e" % (
        "", rel_elem(a, e["arc_logits"], 0.1), rel_elem(a, e["arc_logits"], 0.01)), flush=True)
