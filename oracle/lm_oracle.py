"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone, never by cube_amd).

NumPy restatement of the sub-word -> word pooling of the reference's external-embedding stage,
`LMHelperHF._compute_we` (cube/networks/lm.py:141-197) and the per-layer cut of `apply` (lm.py:199-208), for hidden
states that are given (the transformer itself needs model files that cannot be fetched).  Pinned by golden G12, which
tests/golden/make_golden.py makes by running the reference's own `_compute_we` / `apply` on the synthetic tokenizer /
transformer stand-ins of tests/cases.py.
"""
import numpy as np

START, PAD, END = 0, 1, 2   # lm.py:147-149


def pieces_of(sentences, splitter):
    """lm.py:150-168: -> (input_ids int64 [S, max_len] padded with PAD, word2pieces {(s, w): [(s, pos), ...]})."""
    word2pieces, new_sents = {}, []
    for ii, sent in enumerate(sentences):
        c_sent, pos = [START], 1
        for jj, w in enumerate(sent.words):
            pieces = splitter(w.word)["input_ids"][1:-1]
            word2pieces[(ii, jj)] = []
            for piece in pieces:
                c_sent.append(piece)
                word2pieces[(ii, jj)].append((ii, pos))
                pos += 1
        c_sent.append(END)
        new_sents.append(c_sent)
    max_len = max(len(s) for s in new_sents)
    ids = np.ones((len(new_sents), max_len), dtype=np.int64) * PAD
    for ii, s in enumerate(new_sents):
        ids[ii, :len(s)] = s
    return ids, word2pieces


def pool(we, sentences, word2pieces):
    """lm.py:183-197 on `we` f32 [S, pieces, K*E] (the concatenated hidden states): one vector per word, in order.
    The reference takes `m = we[piece 0]` as a VIEW and then does `m += we[piece z]` for z = 0 .. n-2: the first piece is
    added to itself in place, the last one never enters; `m / n` makes the copy.  Words without pieces: zeros."""
    we = np.array(we, dtype=np.float32, copy=True)
    out = []
    for ii, sent in enumerate(sentences):
        for jj in range(len(sent.words)):
            pieces = word2pieces[ii, jj]
            if pieces:
                m = we[pieces[0][0], pieces[0][1]]
                for zz in range(len(pieces) - 1):
                    m += we[pieces[zz][0], pieces[zz][1]]
                m = m / len(pieces)
            else:
                m = np.zeros(we.shape[-1], dtype=np.float32)
            out.append(m)
    return out


def compute_we(sentences, splitter, encoder):
    """`_compute_we` with the model called in 512-piece windows (lm.py:169-181)."""
    import torch
    ids, word2pieces = pieces_of(sentences, splitter)
    x = torch.from_numpy(ids)
    parts = []
    with torch.no_grad():
        for a in range(0, x.shape[1], 512):
            out = encoder(x[:, a:a + 512], return_dict=True)
            parts.append(torch.cat(out["hidden_states"], dim=-1))
    return pool(torch.cat(parts, dim=1).numpy(), sentences, word2pieces)
