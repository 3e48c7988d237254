"""MI355X-native character-level lemmatizer (inference), SURVEY.md §8f row 2.

Host-side mirror of the reference `Lemmatizer` (cube/networks/lemmatizer.py:20-231) with its
`LemmaDataset` / `Word2TargetCollate` (cube/networks/utils.py:99-122, 153-223): same constructor,
`load`, `forward(X)` and `process(doc, collate, batch_size, num_workers)` and the same results,
with every tensor operation in gfx950 kernels behind the C ABI of include/cube_hip.h.

Per batch of words: the gated char CNN encoder runs once (5 x cube_gemm_bf16 with the gate
epilogue, language embedding folded into per-(language, tap mask) bias tables), then the attention
Conv1d is split: its encoder half is one implicit-conv GEMM per batch, its query half (the query is
constant along the word) is a per-step GEMM of 5 tap projections.
A greedy step is 4 GEMMs (query taps, output heads, LSTM layers 0 and 1) + 3 row kernels, all asynchronous; the host looks at the device's
finished-word counter every `CHECK_EVERY` steps instead of every step (lemmatizer.py:139,161-163).
"""
import ctypes
import os

import numpy as np
import torch

from cube_amd import hip
from cube_amd.networks import packing
from cube_amd.networks.parser import _Buffers


class LemmaDataset:
    """reference: utils.py:99-122.  Inference keeps one example per word (no de-duplication) and
    skips words longer than 25 characters (:109-110)."""

    def __init__(self, document, for_training=True):
        self._examples = []
        lookup = {}
        for sent in document.sentences:
            for w in sent.words:
                if len(w.word) > 25:
                    continue
                key = (w.word, sent.lang_id, w.upos)
                if key not in lookup or for_training is False:
                    lookup[key] = 1
                    self._examples.append({"word": w.word, "upos": w.upos, "lang_id": sent.lang_id, "target": w.lemma})

    def __len__(self):
        return len(self._examples)

    def __getitem__(self, item):
        return self._examples[item]


class Word2TargetCollate:
    """reference: utils.py:153-223 — same keys, dtypes and values (pinned by tests/golden/g9)."""

    def __init__(self, encodings):
        self._encodings = encodings
        self._start_index = len(encodings.char2int)
        self._stop_index = len(encodings.char2int) + 1
        self._eol = len(encodings.char2int)
        self._codes = {}

    def _code(self, ch):
        """(char id, case id) of one character; case: 1 symbol, 2 upper, 3 lower (utils.py:186-196)."""
        hit = self._codes.get(ch)
        if hit is None:
            lo = ch.lower()
            case = 1 if lo == ch.upper() else (2 if lo != ch else 3)
            hit = self._codes[ch] = (self._encodings.char2int.get(lo, self._encodings.char2int["<UNK>"]), case)
        return hit

    def _tables(self, cps):
        """code point -> (char id, case id) arrays, grown on demand; the per-character rule runs once
        per distinct character, the batch is then filled by fancy indexing."""
        top = int(cps.max()) + 1 if len(cps) else 1
        tab = getattr(self, "_cp_tab", None)
        if tab is None or tab.shape[1] < top:
            new = np.full((2, max(top, 256)), -1, dtype=np.int64)
            if tab is not None:
                new[:, :tab.shape[1]] = tab
            self._cp_tab = tab = new
        for cp in np.unique(cps[tab[0, cps] < 0]).tolist():
            tab[:, cp] = self._code(chr(cp))
        return tab

    def _fill(self, texts, first, last):
        """[n, longest + 1 (+1 with a start symbol)] character / case ids: [first,] text, last, zeros."""
        n = len(texts)
        lens = np.fromiter(map(len, texts), dtype=np.int64, count=n)
        off = 0 if first is None else 1
        width = int(lens.max()) + 1 + off
        chars = np.zeros((n, width), dtype=np.int64)
        case = np.zeros((n, width), dtype=np.int64)
        cps = np.frombuffer("".join(texts).encode("utf-32-le", "surrogatepass"), dtype=np.uint32).astype(np.int64)
        tab = self._tables(cps)
        wi = np.repeat(np.arange(n), lens)
        pi = np.arange(len(wi)) - np.repeat(np.cumsum(lens) - lens, lens) + off
        chars[wi, pi] = tab[0, cps]
        case[wi, pi] = tab[1, cps]
        if first is not None:
            chars[:, 0] = first
        chars[np.arange(n), lens + off] = last
        return chars, case

    def collate_fn(self, batch, targets=True):
        enc = self._encodings
        words = [e["word"] for e in batch]
        x_char, x_case = self._fill(words, self._start_index, self._stop_index)
        unk = enc.upos2int["<UNK>"]
        out = {
            "x_char": torch.from_numpy(x_char), "x_case": torch.from_numpy(x_case),
            "x_lang": torch.tensor([e["lang_id"] + 1 for e in batch], dtype=torch.long),
            "x_upos": torch.tensor([enc.upos2int.get(e.get("upos", ""), unk) for e in batch], dtype=torch.long),
        }
        if targets:
            tg = [e["target"] for e in batch]
            y_char, y_case = self._fill(tg, None, self._eol)
            out["y_char"], out["y_case"] = torch.from_numpy(y_char), torch.from_numpy(y_case)
        return out


class Lemmatizer:
    # LSTM cell updates in the epilogue of the gate GEMMs (False / CUBE_LEMMA_FUSED_CELL=0: EPI_F32 + cube_lstm_cell, same bits)
    fused_cell = os.environ.get("CUBE_LEMMA_FUSED_CELL", "1") != "0"
    CHECK_EVERY = 8  # greedy steps between two looks at the finished-word counter

    def __init__(self, config, encodings, language_codes=None):
        if config.decoder_layers != 2:
            raise ValueError("the HIP lemmatizer implements the reference's 2-layer decoder (LemmatizerConfig default)")
        self._config = config
        self._encodings = encodings
        self._num_languages = encodings.num_langs
        self._language_codes = language_codes
        self._eol = len(encodings.char2int)
        self._device = torch.device("cpu")
        self._pk_cpu = self._pk = self._lay = self._buf = None
        self.training = False
        self.precision = "bf16x3"
        self.max_steps = None  # None = the reference's 20 * S failsafe (lemmatizer.py:141); benches cap it

    # ---- reference-compatible module surface ----------------------------------------------
    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, state_dict, strict=True):
        self._pk_cpu, self._lay = packing.pack_lemmatizer(state_dict, self._config, len(self._encodings.char2int))
        self._pk = None
        if self._device.type == "cuda":
            self.to(self._device)

    def load(self, model_path, device="cpu"):
        """reference: lemmatizer.py:176-178 (Lightning checkpoint, only ['state_dict'] is read)."""
        self.load_state_dict(torch.load(model_path, map_location="cpu", weights_only=True)["state_dict"])
        self.to(device)

    def save_packed(self, path):
        """Packed weights as a file for hosts that read them through cube_packfile_* (see `Parser.save_packed`)."""
        if self._pk_cpu is None:
            raise ValueError("load weights first (load / load_state_dict)")
        return packing.save_packed(self._pk_cpu, path, self._lay)

    def to(self, device):
        device = torch.device(device)
        self._device = device
        if device.type != "cuda":
            return self  # the reference defaults to 'cpu'; this implementation has no CPU path
        hip.load()
        if self._pk_cpu is not None:
            if self.precision not in ("bf16x3", "bf16"):
                raise ValueError("precision must be 'bf16x3' or 'bf16'")
            self._split = self.precision == "bf16x3"
            self._pk = packing.to_device(self._pk_cpu, device, with_lo=self._split)
            self._buf = _Buffers(device, self._split)
        return self

    def _require_gpu(self):
        if self._pk is None or self._device.type != "cuda":
            raise hip.CubeHipError("cube_amd Lemmatizer runs only on an MI355X: load weights and call .to('cuda:N') "
                                   "(there is no CPU fallback)")

    # ---- device pipeline -------------------------------------------------------------------
    def _gather(self, dst, ld, col0, width, table, idx, n):
        lo = dst.lo is not None and table.lo is not None
        hip.check(hip.load().cube_gather_rows(dst.hi.data_ptr(), hip.ptr(dst.lo) if lo else None, ld, col0, width,
                                              table.hi.data_ptr(), hip.ptr(table.lo) if lo else None, table.hi.shape[1],
                                              hip.ptr(idx), None, n, hip.stream_handle()), "cube_gather_rows")

    @hip.on_device
    def _encode(self, X):
        """lemmatizer.py:89-124 + the encoder half of the attention Conv1d (modules.py:141,156).
        -> dict(B, S, enc Pair [B*S, enc_w], e_enc f32 [B*S, N])"""
        pk, lay, buf, dev, st = self._pk, self._lay, self._buf, self._device, hip.stream_handle()
        x_char = X["x_char"].cpu().numpy()
        B, S = x_char.shape
        M = B * S
        parts = [x_char.reshape(-1), X["x_case"].cpu().numpy().reshape(-1), np.repeat(X["x_upos"].cpu().numpy(), S),
                 np.repeat(X["x_lang"].cpu().numpy(), S)]
        # from pinned memory: a pageable source would make the copy wait, with the host, for the previous batch's kernels
        ints = torch.from_numpy(np.concatenate(parts).astype(np.int32))
        ints = (ints.pin_memory() if dev.type == "cuda" else ints).to(dev, non_blocking=True)
        char_i, case_i, upos_i, lang_i = (ints[k * M:(k + 1) * M] for k in range(4))
        self._live = [ints]  # per-batch device inputs: referenced until the next batch replaces them
        xe0 = buf.get("xe0", M, lay.xe0_w)
        self._gather(xe0, lay.xe0_w, 0, packing.p8(lay.ce), pk["char_emb"], char_i, M)
        self._gather(xe0, lay.xe0_w, lay.xe0_case, lay.case, pk["case_emb"], case_i, M)
        self._gather(xe0, lay.xe0_w, lay.xe0_upos, packing.p8(lay.ue), pk["upos_emb"], upos_i, M)
        enc = buf.get("enc", M, lay.enc_w)
        self._gather(enc, lay.enc_w, lay.enc_case, lay.case, pk["case_emb"], case_i, M)
        self._gather(enc, lay.enc_w, lay.enc_upos, packing.p8(lay.ue), pk["upos_emb"], upos_i, M)
        self._gather(enc, lay.enc_w, lay.enc_lang, packing.p8(lay.le), pk["lang_emb"], lang_i, M)
        ga, gb = buf.get("ga", M, lay.half), buf.get("gb", M, lay.half)
        res = buf.get("res", M, lay.half, torch.float32)
        src, src_ld = xe0, lay.xe0_w
        n = len(pk["convs"])
        for l, cv in enumerate(pk["convs"]):
            last = l == n - 1
            dst = None if last else (ga if l % 2 == 0 else gb)
            hip.gemm(src, src_ld, M, cv["Kc"], cv["W"], cv["N"], cv["bias"], hip.EPI_GATE, out0=dst, ldo0=lay.half, col0=0,
                     taps=cv["taps"], seq_len=S, n_valid=cv["half"], res=res, ld_res=lay.half, res_mode=0 if l == 0 else 1,
                     out1=enc if last else None, ldo1=lay.enc_w, col1=0, bias_tab=cv["bias_tab"], row_group=lang_i, stream=st)
            src, src_ld = dst, lay.half
        ae = pk["att_enc"]
        e_enc = buf.get("e_enc", M, ae["N"], torch.float32)
        hip.gemm(enc, lay.enc_w, M, ae["Kc"], ae["W"], ae["N"], ae["bias"], hip.EPI_F32, out0=e_enc, ldo0=ae["N"],
                 n_valid=ae["N"], taps=ae["taps"], seq_len=S, stream=st)
        return dict(B=B, S=S, enc=enc, e_enc=e_enc, ld_e=ae["N"], _ints=ints)

    @hip.on_device
    def _decode(self, ed, max_steps=None, keep_logits=False, defer=False):
        """Greedy decoding (lemmatizer.py:126-169).  -> (out_char i32 [B, n], out_case i32 [B, n], n, logits)
        with n = the number of steps the reference would have run.  defer: everything is enqueued, the arg-max arrays are
        written to pinned memory by a kernel (cube_copy_to_pinned) and a handle for `_decode_finish` is returned instead, so
        that the caller can featurise and enqueue the NEXT batch before it waits for this one."""
        lib, pk, lay, buf, dev, st = hip.load(), self._pk, self._lay, self._buf, self._device, hip.stream_handle()
        B, S = ed["B"], ed["S"]
        limit = S * 20 if max_steps is None else max_steps
        D, ld_x = lay.D, lay.xd_w
        fused = self.fused_cell
        # fused: two decoder-state buffers taken in turn (X[cur] holds [ctx | prev char | h0] of the step that runs next and
        # h1, c1 of the step before; a cell update in the GEMM epilogue may not write h into the columns other tiles read)
        xd2 = buf.get("xd", 2 * B, ld_x)
        xd2.hi[:2 * B * ld_x].zero_()
        if xd2.lo is not None:
            xd2.lo[:2 * B * ld_x].zero_()
        X = [hip.Pair(xd2.hi[k * B * ld_x:(k + 1) * B * ld_x], None if xd2.lo is None else xd2.lo[k * B * ld_x:(k + 1) * B * ld_x])
             for k in range(2 if fused else 1)]
        xd = X[0]
        cs = buf.get("cells", 2 * B, D, torch.float32)
        cs[:2 * B * D].zero_()
        c0, c1 = cs[:B * D], cs[B * D:2 * B * D]
        l0, l1, qq, hh = pk["lstm0"], pk["lstm1"], pk["query"], pk["heads"]
        f0, f1 = pk["lstm0f"], pk["lstm1f"]
        gates = None if fused else buf.get("gates", B, l0["N"], torch.float32)
        ld_q = lay.q_w
        q = buf.get("q", B, ld_q, torch.float32)
        out = torch.zeros((2, B, max(limit, 1)), dtype=torch.int32, device=dev)
        flags = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        logits = torch.zeros((limit, B, lay.n_char + lay.n_case), dtype=torch.float32, device=dev) if keep_logits else None
        self._live += [out, flags, logits]
        lo = hip.ptr

        def lstm(cur):
            """Both decoder layers for one time step (lemmatizer.py:131,149); the input row is X[cur]."""
            if fused:   # the cell update is the epilogue of the gate GEMM (CUBE_EPI_LSTM): h0 -> X[1-cur], h1 / c1 -> X[cur]
                xi, xo = X[cur], X[1 - cur]
                hip.gemm(xi, ld_x, B, f0["K"], f0["W"], f0["N"], f0["bias"], hip.EPI_LSTM, out0=xo, ldo0=ld_x, col0=lay.xd_h0,
                         n_valid=D, res=c0, ld_res=D, stream=st)
                hip.gemm(xo, ld_x, B, f1["K"], f1["W"], f1["N"], f1["bias"], hip.EPI_LSTM, out0=xi, ldo0=ld_x, col0=lay.xd_h1,
                         n_valid=D, res=c1, ld_res=D, out1=xi, ldo1=ld_x, col1=lay.xd_c1, a_col=lay.xd_h0, stream=st)
                return
            hip.gemm(xd, ld_x, B, l0["K"], l0["W"], l0["N"], l0["bias"], hip.EPI_F32, out0=gates, ldo0=l0["N"],
                     n_valid=l0["N"], stream=st)
            hip.check(lib.cube_lstm_cell(gates.data_ptr(), l0["N"], D, c0.data_ptr(), B, xd.hi.data_ptr(), lo(xd.lo), ld_x,
                                         lay.xd_h0, None, None, 0, st), "cube_lstm_cell")
            hip.gemm(xd, ld_x, B, l1["K"], l1["W"], l1["N"], l1["bias"], hip.EPI_F32, out0=gates, ldo0=l1["N"],
                     n_valid=l1["N"], a_col=lay.xd_h0, stream=st)
            hip.check(lib.cube_lstm_cell(gates.data_ptr(), l1["N"], D, c1.data_ptr(), B, xd.hi.data_ptr(), lo(xd.lo), ld_x,
                                         lay.xd_h1, xd.hi.data_ptr(), lo(xd.lo), lay.xd_c1, st), "cube_lstm_cell")

        a = hip.LemmaAttendArgs()
        a.q, a.ld_q, a.taps, a.P, a.ld_tap = q.data_ptr(), ld_q, pk["att_enc"]["taps"], lay.P, lay.ld_tap
        a.col_char, a.n_char, a.col_case, a.n_case = lay.q_char, lay.n_char, lay.q_case, lay.n_case
        a.v, a.e_enc, a.ld_e = pk["att_v"].data_ptr(), ed["e_enc"].data_ptr(), ed["ld_e"]
        a.enc, a.enc_lo, a.ld_enc, a.C = ed["enc"].hi.data_ptr(), lo(ed["enc"].lo), lay.enc_w, lay.enc_c
        a.B, a.S = B, S
        a.char_emb, a.char_emb_lo = pk["char_emb"].hi.data_ptr(), lo(pk["char_emb"].lo)
        a.ld_emb, a.E = pk["char_emb"].hi.shape[1], packing.p8(lay.ce)
        a.ld_x, a.col_ctx, a.col_emb = ld_x, 0, lay.xd_emb
        a.out_char, a.out_case, a.ld_out = out[0].data_ptr(), out[1].data_ptr(), out.shape[2]
        a.eol, a.done, a.n_done = self._eol, flags.data_ptr(), flags[B:].data_ptr()

        # start frame through the decoder with a zero state (lemmatizer.py:128-131)
        cur = 0
        self._gather(X[cur], ld_x, 0, lay.xd_h0, pk["start"], None, B)
        lstm(cur)
        steps = 0
        for s in range(limit + 1):
            xs = X[cur]         # h1 / c1 of the step that has just run
            if s < limit:   # tap projections of the query (cell state c1) for this step's attention
                hip.gemm(xs, ld_x, B, qq["K"], qq["W"], qq["N"], qq["bias"], hip.EPI_F32, out0=q, ldo0=ld_q, n_valid=qq["N"],
                         a_col=lay.xd_c1, stream=st)
            if s > 0:       # output heads on h1 of the previous step
                hip.gemm(xs, ld_x, B, hh["K"], hh["W"], hh["N"], hh["bias"], hip.EPI_F32, out0=q, ldo0=ld_q, col0=lay.q_char,
                         n_valid=hh["N"], a_col=lay.xd_h1, stream=st)
            if keep_logits and s > 0:
                logits[s - 1].copy_(q[:B * ld_q].view(B, ld_q)[:, lay.q_char:lay.q_char + lay.n_char + lay.n_case])
            if fused:
                cur = 1 - cur   # context and the chosen character's embedding go into the NEXT step's input row
            a.x, a.x_lo = X[cur].hi.data_ptr(), lo(X[cur].lo)
            a.step, a.attend = s, int(s < limit)
            hip.check(lib.cube_lemma_attend(ctypes.byref(a), st), "cube_lemma_attend")
            steps = s
            if s == limit:
                break
            if s > 0 and s % self.CHECK_EVERY == 0 and int(flags[B].item()) == B:
                break
            lstm(cur)
        if defer:
            n_int = out.numel()
            ring = self.__dict__.setdefault("_pinned_ring", [])
            slot = self.__dict__.get("_pinned_next", 0)
            self._pinned_next = (slot + 1) % 3     # at most two handles are outstanding at any time (process loops below)
            while len(ring) <= slot:
                ring.append(torch.empty(0, dtype=torch.int32))
            if ring[slot].numel() < n_int:
                ring[slot] = torch.empty(n_int, dtype=torch.int32, pin_memory=True)
            hip.check(lib.cube_copy_to_pinned(out.data_ptr(), ring[slot].data_ptr(), 4 * n_int, st), "cube_copy_to_pinned")
            ev = torch.cuda.Event()
            ev.record()
            return dict(host=ring[slot], shape=tuple(out.shape), steps=steps, event=ev, out=out, logits=logits)
        res = out[:, :, :max(steps, 1)].cpu().numpy()
        return self._trim(res, steps, logits)

    def _decode_finish(self, h):
        """Wait for a deferred `_decode` and trim its arrays as the synchronous form does."""
        h["event"].synchronize()
        steps = h["steps"]
        res = h["host"][:int(np.prod(h["shape"]))].numpy().reshape(h["shape"])[:, :, :max(steps, 1)].copy()
        h["out"] = None
        return self._trim(res, steps, h["logits"])

    def _trim(self, res, steps, logits):
        oc, ok = res[0][:, :steps], res[1][:, :steps]
        # the reference stops right after the step at which the last word emitted EOL (lemmatizer.py:139)
        n = steps
        if steps > 0:
            is_eol = oc == self._eol
            if is_eol.any(axis=1).all():
                n = int(is_eol.argmax(axis=1).max()) + 1
        return oc[:, :n], ok[:, :n], n, (None if logits is None else logits[:n])

    @hip.on_device
    def forward(self, X):
        """-> (char logits [B, n, C+2], case logits [B, n, 4]) of the greedy decode, as the reference's
        inference branch returns them (lemmatizer.py:171).  Teacher forcing ('y_char' in X) is training-only
        and not implemented."""
        self._require_gpu()
        if "y_char" in X:
            raise NotImplementedError("teacher-forced forward is a training path; delete y_char / y_case as process() does")
        oc, ok, n, logits = self._decode(self._encode(X), keep_logits=True)
        logits = logits.permute(1, 0, 2)
        return logits[:, :, :self._lay.n_char], logits[:, :, self._lay.n_char:]

    @hip.on_device
    def decode_batch(self, X):
        """-> (chars int32 [B, n], cases int32 [B, n]) arg-max sequences (NumPy), no logits kept."""
        self._require_gpu()
        oc, ok, _, _ = self._decode(self._encode(X), max_steps=self.max_steps)
        return oc, ok

    @hip.on_device
    def _decode_batches(self, batches, collate):
        """batches: iterable of lists of items for `collate`.  Yields (index, lemma strings) per batch, in order, with the
        host work of neighbouring batches (collate of the next, strings of the previous) under the device's decode."""
        def finish(k, h):
            oc, ok, _, _ = self._decode_finish(h)
            return k, self._strings(oc, ok)

        pending = None
        for k, items in enumerate(batches):
            batch = collate.collate_fn(items, targets=False) if isinstance(collate, Word2TargetCollate) else collate.collate_fn(items)
            batch.pop("y_char", None)
            batch.pop("y_case", None)
            h = self._decode(self._encode(batch), max_steps=self.max_steps, defer=True)
            if pending is not None:
                yield finish(*pending)
            pending = (k, h)
        if pending is not None:
            yield finish(*pending)

    def _strings(self, oc, ok):
        """arg-max ids -> lemma strings (lemmatizer.py:210-224): characters up to the first EOL, the case
        head upper-/lower-cases each one.  Rows whose characters all map to single code points (all but
        '<PAD>', '<UNK>' and letters whose upper case is a sequence) are assembled as one UTF-32 buffer."""
        tabs = getattr(self, "_char_tabs", None)
        if tabs is None:
            chars = self._encodings.characters
            obj = tuple(np.array([f(c) for c in chars] + [""], dtype=object) for f in (str, str.upper, str.lower))
            single = np.array([all(len(t[i]) == 1 for t in obj) for i in range(len(chars))] + [True])
            cps = tuple(np.array([ord(x) if len(x) == 1 else 0 for x in t.tolist()], dtype=np.uint32) for t in obj)
            tabs = self._char_tabs = (obj, cps, single)
        obj, cps, single = tabs
        pad = len(single) - 1
        B, n = oc.shape
        is_eol = oc == self._eol
        end = np.where(is_eol.any(axis=1), is_eol.argmax(axis=1), n) if n else np.zeros(B, dtype=np.int64)
        live = np.arange(n)[None, :] < end[:, None]
        if (oc[live] >= pad).any():
            raise IndexError("list index out of range")  # the stop index has no character (lemmatizer.py:217)
        idx = np.where(live, oc, pad)
        fast = single[idx].all(axis=1)
        out = [None] * B
        rows = np.where(fast)[0]
        if len(rows):
            i, k, lv = idx[rows], ok[rows], live[rows]
            cp = np.where(k == 2, cps[1][i], np.where(k == 3, cps[2][i], cps[0][i]))
            text = cp[lv].astype("<u4").tobytes().decode("utf-32-le", "surrogatepass")
            pos = 0
            for r, e in zip(rows.tolist(), end[rows].tolist()):
                out[r] = text[pos:pos + e]
                pos += e
        for r in np.where(~fast)[0].tolist():
            i, k = idx[r, :end[r]], ok[r, :end[r]]
            out[r] = "".join(np.where(k == 2, obj[1][i], np.where(k == 3, obj[2][i], obj[0][i])).tolist())
        return out

    @hip.on_device
    def process_bucketed(self, doc, collate, max_words=8192):
        """Throughput mode, no counterpart in the reference: every distinct (word, UPOS, language) is
        lemmatised ONCE, in batches of words of the SAME length (S = length + 2, so no pad positions
        take part in the convolutions or the attention and a word's lemma does not depend on its
        batch).  Lemmas therefore differ from `process` exactly where the reference's own results
        depend on batch composition.  Words longer than 25 characters keep their lemma (the
        reference's dataset skips them, utils.py:109-110, and then mis-assigns, lemmatizer.py:225-229)."""
        self._require_gpu()
        from cube_amd import host
        hm = host.load()
        fast = None
        if hm is not None and isinstance(doc.sentences, list):
            # the two passes over every word run before / after all device work: one C pass each (csrc/host_collate.c)
            try:
                fast = hm.lemma_keys(doc.sentences, 25)
            except TypeError:
                fast = None   # unusual objects: the Python pass below says what is wrong, or handles them
        if fast is not None:
            k_word, k_upos, k_lang, lens, widx = fast
            lens = lens.astype(np.int64)
            n_keys = len(k_word)

            def item(i):
                return {"word": k_word[i], "upos": k_upos[i], "lang_id": int(k_lang[i]), "target": ""}
        else:
            keys, order, targets, widx = {}, [], [], []   # key -> index, keys by index, the words to fill, their key's index
            for sent in doc.sentences:
                lang = sent.lang_id
                for w in sent.words:
                    word = w.word
                    if len(word) <= 25:
                        k = (word, w.upos, lang)
                        i = keys.get(k)
                        if i is None:
                            i = keys[k] = len(order)
                            order.append(k)
                        targets.append(w)
                        widx.append(i)
            lens = np.fromiter((len(k[0]) for k in order), dtype=np.int64, count=len(order))
            n_keys = len(order)

            def item(i):
                return {"word": order[i][0], "upos": order[i][1], "lang_id": order[i][2], "target": ""}
        perm = np.argsort(lens, kind="stable")
        parts = []   # lists of key indices: equal word length, at most max_words each
        cuts = np.flatnonzero(np.diff(lens[perm])) + 1
        for seg in np.split(perm, cuts):
            parts += [seg[s:s + max_words].tolist() for s in range(0, len(seg), max_words)]
        lemma_of = [None] * n_keys
        for b, lemmas in self._decode_batches(([item(i) for i in part] for part in parts), collate):
            for i, lemma in zip(parts[b], lemmas):
                lemma_of[i] = lemma
        if fast is not None:
            hm.assign_lemmas(doc.sentences, widx, lemma_of, 25)
        else:
            for w, i in zip(targets, widx):
                w.lemma = lemma_of[i]
        return doc

    @hip.on_device
    def process(self, doc, collate, batch_size=4, num_workers=4, merge=True, max_words=8192):
        """reference: lemmatizer.py:185-231.  `num_workers` is accepted and ignored.
        merge (no counterpart in the reference, results unchanged): the reference's in-order batches of
        `batch_size` words whose longest word has the same length (same S, hence the same pad positions
        for every word in them) run together as one device batch of up to `max_words` words; words do not
        interact otherwise (`sharding.merged_reference_batches`)."""
        self._require_gpu()
        dataset = LemmaDataset(doc, for_training=False)
        n = len(dataset)
        starts = list(range(0, n, batch_size))
        if merge and len(starts) > 1:
            from cube_amd.sharding import merged_reference_batches
            keys = [max(len(dataset[i]["word"]) for i in range(a, min(a + batch_size, n))) for a in starts]
            groups = merged_reference_batches(keys, batch_size, max_words)
            device_batches = [[i for r in g for i in range(starts[r], min(starts[r] + batch_size, n))] for g in groups]
        else:
            device_batches = [list(range(a, min(a + batch_size, n))) for a in starts]
        all_lemmas = [None] * n
        for b, lemmas in self._decode_batches(([dataset[i] for i in ids] for ids in device_batches), collate):
            for i, lemma in zip(device_batches[b], lemmas):
                all_lemmas[i] = lemma
        i = 0
        for sent in doc.sentences:
            for w in sent.words:
                w.lemma = all_lemmas[i]  # IndexError when a word was skipped (> 25 characters), as in the reference
                i += 1
        return doc
