"""Lemmatizer throughput on one MI355X (SURVEY.md §8f-2 / BASELINE.json configs[4] stage).

    python scripts/bench_lemma.py [--words 4096] [--steps 32] [--cpu-words 64]

Synthetic random-init model (cube_amd.synth), words of the §8d length distribution.  A random
decoder does not stop at a sensible length, so the decode is timed for a FIXED number of greedy
steps (default 32: > 99 % of English lemmas are shorter) and reported as words/s at that length and
as word-steps/s; the encoder is timed separately.  The CPU figure is the oracle (PyTorch-CPU fp32
restatement of the reference) on a small sample of the same batch, same number of steps."""
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
from cube_amd.io_utils.config import LemmatizerConfig  # noqa: E402
from cube_amd.networks.lemmatizer import LemmaDataset, Lemmatizer, Word2TargetCollate  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--words", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--cpu-words", type=int, default=64)
    ap.add_argument("--precision", default="bf16x3")
    a = ap.parse_args()
    cfg = LemmatizerConfig()
    enc = synth.make_encodings()
    sd = synth.make_lemmatizer_state_dict(cfg, enc)
    n_sent = (a.words + 31) // 32
    doc = synth.make_document(enc, [32] * n_sent, seed=5, ext_word_emb=())
    rng = np.random.default_rng(6)
    for s in doc.sentences:
        for w in s.words:
            w.upos = enc.upos_list[int(rng.integers(2, len(enc.upos_list)))]
    ds = LemmaDataset(doc, for_training=False)
    col = Word2TargetCollate(enc)
    X = col.collate_fn([ds[i] for i in range(a.words)], targets=False)
    B, S = X["x_char"].shape
    m = Lemmatizer(cfg, enc)
    m.precision = a.precision
    m.load_state_dict(sd)
    m.to("cuda:0")
    m.CHECK_EVERY = 1 << 30  # fixed step count: never look at the finished-word counter
    for _ in range(2):
        m._decode(m._encode(X), max_steps=a.steps)
    torch.cuda.synchronize()
    enc_ms, dec_ms = [], []
    for _ in range(a.reps):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        ed = m._encode(X)
        e1.record()
        m._decode(ed, max_steps=a.steps)
        e2.record()
        torch.cuda.synchronize()
        enc_ms.append(e0.elapsed_time(e1))
        dec_ms.append(e1.elapsed_time(e2))
    t_enc, t_dec = float(np.median(enc_ms)), float(np.median(dec_ms))
    out = dict(workload="lemmatizer B=%d words, S=%d, %d greedy steps" % (B, S, a.steps), precision=a.precision,
               encode_ms=round(t_enc, 3), decode_ms=round(t_dec, 3), ms_per_step=round(t_dec / a.steps, 4),
               words_per_s=round(B / ((t_enc + t_dec) * 1e-3), 1), word_steps_per_s=round(B * a.steps / (t_dec * 1e-3), 1))
    if a.cpu_words > 0:
        from oracle import lemmatizer_oracle as lo
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        orc = lo.OracleLemmatizer(cfg, enc, sd)
        Xc = lo.collate(lo.dataset(doc)[:a.cpu_words], enc)
        orc.forward(Xc, max_steps=2)
        t0 = time.perf_counter()
        orc.forward(Xc, max_steps=a.steps)
        dt = time.perf_counter() - t0
        out["cpu_baseline"] = dict(kind="port", words=a.cpu_words, threads=torch.get_num_threads(),
                                   words_per_s=round(a.cpu_words / dt, 1), note="oracle, same steps; stops early if every word emits EOL")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
