"""MI355X-native tokenizer / sentence-splitter network (inference), SURVEY.md §8f row 4.

Host-side mirror of the reference `Tokenizer` (cube/networks/tokenizer.py:25-127, 220-304): same
constructor, `load`, `to`, `forward(batch)` and `process(raw_text, collate, batch_size, num_workers,
lang_id)`, with every tensor operation in the gfx950 kernels of include/cube_hip.h — the same
WordGram character CNN, external-embedding MLPs and gated Conv1d stack as the parser (the stack
here feeds every layer the running SUM of all previous gates, tokenizer.py:116-122: gate epilogue
`res_mode` 3), then `Linear -> 5` labels per sub-word piece.

The sub-word collate (`TokenCollateHF` / `TokenCollateFTLanguasito`, utils_tokenizer.py) runs a
HuggingFace / fastText model that this package does not re-implement: `process` takes any collate
object with the reference's `get_tokens` / `collate_fn` (the reference's own classes work unchanged)
and batches in their dict format.
"""
import numpy as np
import torch

from cube_amd import hip
from cube_amd.io_utils.objects import Document, Sentence, Word
from cube_amd.networks import packing
from cube_amd.networks.packing import p32
from cube_amd.networks.parser import _HipNet


class Tokenizer(_HipNet):
    _root = 0
    _pack = staticmethod(packing.pack_tokenizer)

    def __init__(self, config, encodings, language_codes=None, ext_word_emb=0, max_seq_len=-1):
        super().__init__(config, encodings, language_codes, ext_word_emb)
        self._char_layers = 3  # WordGram default (tokenizer.py:49, modules.py:426)
        self._max_seq_len = max_seq_len

    @hip.on_device
    def _prepare(self, X):
        """reference-format batch dict (TokenCollate*.collate_fn) -> device index arrays + embeddings."""
        self._require_gpu()
        dev = self._device
        x_word = X["x_word_char"].cpu().numpy()
        x_case = X["x_word_case"].cpu().numpy()
        word_len = X["x_word_len"].cpu().numpy().astype(np.int64)
        sl = X["x_sent_len"].cpu().numpy().astype(np.int64)
        B, T = X["x_input"][0].shape[0], X["x_input"][0].shape[1]
        N, W = x_word.shape
        assert int(sl.sum()) == N and len(sl) == B and (T >= sl.max())
        ch = self._char_layout(word_len, W)
        order = ch["order"]
        char_idx = np.concatenate([x_word[order[s:e], :q].reshape(-1) for s, e, q in ch["gather"]])
        case_idx = np.concatenate([x_case[order[s:e], :q].reshape(-1) for s, e, q in ch["gather"]])
        l0 = self._pk.get("char_l0")
        if l0 is not None:
            char_idx, case_idx = char_idx * l0["n_case"] + case_idx, np.zeros(0, dtype=np.int64)
        sb = np.repeat(np.arange(B), sl)
        si = np.arange(N) - np.repeat(np.cumsum(sl) - sl, sl)
        parts = [char_idx, case_idx, X["x_word_lang"].cpu().numpy()[order], word_len[order], (sb * T + si)[order],
                 X["x_input_spa"].cpu().numpy().reshape(-1), np.repeat(X["x_lang"].cpu().numpy(), T), ch["word_rows"].reshape(-1)]
        sizes = [p.size for p in parts]
        offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sizes])])
        packed = np.zeros(int(offs[-1]), dtype=np.int32)
        for p, o, s in zip(parts, offs[:-1], sizes):
            packed[o:o + s] = p
        ints = torch.from_numpy(packed).to(dev, non_blocking=True)
        names = ("char_idx", "case_idx", "lang_word", "word_len", "rowmap_word", "spa_idx", "lang_rows", "word_rows")
        bt = {n: ints[o:o + s] for n, o, s in zip(names, offs[:-1], sizes)}
        embs = []
        for k, e in enumerate(X["x_input"]):
            ek = self._pk["ext"][k]
            e = e.to(dev, non_blocking=True).to(torch.float32).reshape(B * T, -1)
            pad = torch.zeros((B * T, ek["Epad"]), dtype=torch.float32, device=dev)
            pad[:, :ek["E"]] = e
            embs.append(hip.Pair.split(pad, with_lo=self._split))
        bt.update(B=B, T=T, N=N, W=W, embs=embs, conv_groups=ch["groups"], char_rows=ch["rows"], _ints=ints)
        return bt

    def _logits(self, bt):
        lib, pk, lay, buf, st = hip.load(), self._pk, self._lay, self._buf, hip.stream_handle()
        lo = hip.ptr
        B, T, N = bt["B"], bt["T"], bt["N"]
        BT = B * T
        wh, P = lay.word_half, lay.P
        self._last_bt = bt  # per-batch device inputs stay referenced until the next batch
        pool, pool_w = self._char_level(bt)
        # XW0 = [ext | char | spa]; pad pieces keep a zero character vector (unpack, utils.py:17-32)
        xw0 = buf.get("xw0", BT, lay.xw0_w)
        if N != BT:
            self._zero(xw0, BT * lay.xw0_w)
        po = pk["char_pre_out"]
        hip.gemm(pool, pool_w, N, po["K"], po["W"], po["N"], po["bias"], hip.EPI_BF16_TANH, out0=xw0, ldo0=lay.xw0_w,
                 col0=lay.xw0_char, n_valid=lay.char_half, rowmap=bt["rowmap_word"], stream=st)
        self._gather(xw0, lay.xw0_w, lay.xw0_spa, lay.spa, pk["spa_emb"], bt["spa_idx"], None, BT)
        ext_h = buf.get("ext_h", BT, pk["ext"][0]["h1"]["N"])
        ld_acc = 8 * ((P + 7) // 8)
        ext_acc = buf.get("ext_acc", BT, ld_acc, torch.float32)
        for k, ek in enumerate(pk["ext"]):
            h1, h2 = ek["h1"], ek["h2"]
            hip.gemm(bt["embs"][k], ek["Epad"], BT, h1["K"], h1["W"], h1["N"], h1["bias"], hip.EPI_BF16_TANH,
                     out0=ext_h, ldo0=h1["N"], n_valid=h1["N"], stream=st)
            hip.gemm(ext_h, h1["N"], BT, h2["K"], h2["W"], h2["N"], h2["bias"], hip.EPI_F32, out0=ext_acc, ldo0=ld_acc,
                     n_valid=P, accumulate=1 if k > 0 else 0, stream=st)
        hip.check(lib.cube_ext_finish(ext_acc.data_ptr(), ld_acc, B, T, 0, P, len(pk["ext"]), 1, 0, xw0.hi.data_ptr(), lo(xw0.lo),
                                      lay.xw0_w, 0, None, None, 0, 0, st), "cube_ext_finish")
        # gated Conv1d stack; every layer reads the running sum of the gates before it
        xa, xb = buf.get("xwa", BT, lay.xw_w), buf.get("xwb", BT, lay.xw_w)
        res = buf.get("resw", BT, wh, torch.float32)
        fin = buf.get("fin", BT, lay.fin_w)
        self._gather(fin, lay.fin_w, wh, packing.p8(lay.word_lang), pk["lang_emb"], bt["lang_rows"], None, BT)
        n = len(pk["word_convs"])
        src, src_ld = xw0, lay.xw0_w
        for l, cv in enumerate(pk["word_convs"]):
            last = l == n - 1
            dst = None if last else (xa if l % 2 == 0 else xb)
            hip.gemm(src, src_ld, BT, cv["Kc"], cv["W"], cv["N"], cv["bias"], hip.EPI_GATE, out0=dst, ldo0=lay.xw_w, col0=0,
                     taps=cv["taps"], seq_len=T, n_valid=cv["half"], res=res, ld_res=wh,
                     res_mode=0 if l == 0 else (1 if last else 3), out1=fin if last else None, ldo1=lay.fin_w, col1=0,
                     bias_tab=cv["bias_tab"], row_group=bt["lang_rows"], stream=st)
            src, src_ld = dst, lay.xw_w
        q = pk["output"]
        logits = bt["_logits"] = torch.empty((BT, q["n_valid"]), dtype=torch.float32, device=self._device)
        hip.gemm(fin, lay.fin_w, BT, q["K"], q["W"], q["N"], q["bias"], hip.EPI_F32, out0=logits, ldo0=q["n_valid"],
                 n_valid=q["n_valid"], stream=st)
        return logits.view(B, T, q["n_valid"])

    @hip.on_device
    def forward(self, batch):
        """-> logits f32 [B, T, 5] on the device (tokenizer.py:70-127)."""
        return self._logits(self._prepare(batch))

    @hip.on_device
    def process(self, raw_text, collate, batch_size=32, num_workers=4, lang_id=0, dataset=None):
        """reference: tokenizer.py:224-304.  `collate` supplies `get_tokens(text)` and `collate_fn(examples)`
        (the reference's TokenCollate classes); `dataset` defaults to the reference's TokenDatasetLive
        windows (300 pieces + 100 of context each side, utils_tokenizer.py:58-112) built here."""
        self._require_gpu()
        raw_text = raw_text.replace("\n", " ").replace("\r", " ")
        new_text = raw_text.replace("  ", " ")
        while new_text != raw_text:
            raw_text = new_text
            new_text = raw_text.replace("  ", " ")
        if dataset is None:
            dataset = token_windows(raw_text, collate.get_tokens)
        collate._lang_id = lang_id
        toks, preds = [], []
        for s in range(0, len(dataset), batch_size):
            batch = collate.collate_fn([dataset[i] for i in range(s, min(s + batch_size, len(dataset)))])
            y = torch.argmax(self.forward(batch), dim=-1).cpu().numpy()
            y_offset, y_len = batch["y_offset"].cpu().numpy(), batch["y_len"].cpu().numpy()
            for ii in range(len(y_len)):
                toks.extend(batch["x_text"][ii][:y_len[ii]])
                preds.extend(y[ii, y_offset[ii]:y_offset[ii] + y_len[ii]].tolist())
        return labels_to_document(toks, preds, lang_id)


def token_windows(raw_text, pretokenize, seq_len=300, overlap=100):
    """TokenDatasetLive (utils_tokenizer.py:58-112): windows of `seq_len` pieces with up to `overlap`
    pieces of context before and after; each example {'prev', 'main', 'next'} of (pieces[, ids], spaces)."""
    if raw_text[-1:] != " ":
        raw_text += " "
    toks = pretokenize(raw_text)
    n = len(toks[0])
    out = []
    for i in range((n + seq_len - 1) // seq_len):
        start, stop = i * seq_len, min(i * seq_len + seq_len, n)
        left, right = max(0, start - overlap), min(n, stop + overlap)
        out.append({"prev": tuple(t[left:start] for t in toks), "main": tuple(t[start:stop] for t in toks),
                    "next": tuple(t[stop:right] for t in toks)})
    return out


def labels_to_document(toks, preds, lang_id):
    """tokenizer.py:258-304: labels 2 / 3 / 4 close a token / multi-word token / token + sentence."""
    d = Document()
    sent, tok = [], ""

    def close_sentence():
        if sent:
            seq = []
            for cnt, (w, m) in enumerate(sent, 1):
                seq.append(Word(cnt, w, "_", "_", "_", "_", 0, "_", "_", "_"))
                if m:
                    seq[-1].space_after += ";compund"
            d.sentences.append(Sentence(sequence=seq, lang_id=lang_id))
        del sent[:]

    for pred, text in zip(preds, toks):
        tok += text.replace("▁", "")
        if pred in (2, 3, 4):
            if tok.strip() != "":
                sent.append((tok, pred == 3))
            tok = ""
        if pred == 4:
            close_sentence()
    if tok.strip() != "":
        sent.append((tok, False))
    close_sentence()
    return d
