"""Tokenizer network throughput on one MI355X.

    python scripts/bench_tokenizer.py [--windows 32] [--pieces 500] [--ext 13x768]

One step = `Tokenizer.forward` on a device-resident batch of `windows` examples of `pieces` sub-word
pieces (the reference's live windows are 300 pieces + 100 of context each side, utils_tokenizer.py:97)
with K x E "LM" values per piece (13 x 768 = xlm-roberta-base hidden states); synthetic pieces and
embeddings, random-init weights.  The sub-word LM itself is not part of this package."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nlp-cube_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from cube_amd import synth  # noqa: E402
from cube_amd.io_utils.config import TokenizerConfig  # noqa: E402
from cube_amd.networks.tokenizer import Tokenizer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--windows", type=int, default=32)
    ap.add_argument("--pieces", type=int, default=500)
    ap.add_argument("--ext", default="13x768")
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--cpu", action="store_true", help="also time the oracle on one batch")
    a = ap.parse_args()
    k, e = a.ext.lower().split("x")
    ext = [int(e)] * int(k)
    cfg, enc = TokenizerConfig(), synth.make_encodings()
    sd = synth.make_tokenizer_state_dict(cfg, enc, ext)
    m = Tokenizer(cfg, enc, ext_word_emb=ext)
    m.load_state_dict(sd)
    m.to("cuda:0")
    X = synth.make_token_batch(enc, [a.pieces] * a.windows, ext, seed=1)
    bt = m._prepare(X)
    for _ in range(3):
        m._logits(bt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = m._logits(bt)
    y.argmax(-1).cpu()
    dt = (time.perf_counter() - t0) / a.steps
    out = dict(workload="tokenizer forward, %d windows x %d pieces, ext %s" % (a.windows, a.pieces, a.ext),
               ms_per_step=round(dt * 1e3, 3), pieces_per_s=round(a.windows * a.pieces / dt, 1))
    if a.cpu:
        from oracle import tokenizer_oracle as to
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        net = to.OracleTokenizer(cfg, enc, sd, ext)
        t0 = time.perf_counter()
        net.forward(X)
        out["cpu_baseline"] = dict(kind="port", threads=torch.get_num_threads(),
                                   pieces_per_s=round(a.windows * a.pieces / (time.perf_counter() - t0), 1))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
