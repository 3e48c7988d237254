"""Host-side batching for the parser/tagger path.

Mirror of the reference's `MorphoDataset` / `MorphoCollate` / `unpack`
(cube/networks/utils.py:17-32, 88-96, 226-330): same class names, same `collate_fn(batch)`
result keys, dtypes and values (pinned by tests/golden/g1_collate.npz), built with one pass
over the characters instead of the reference's per-element NumPy stores.
`ChuLiuEdmondsDecoder.decode` keeps the reference signature but runs `cube_mst_decode` on
the GPU.
"""
import numpy as np
import torch

from cube_amd import hip, host


class MorphoDataset:
    """reference: utils.py:88-96"""

    def __init__(self, document):
        self._document = document

    def __len__(self):
        return len(self._document.sentences)

    def __getitem__(self, item):
        return self._document.sentences[item]


class MorphoCollate:
    """reference: utils.py:226-330 (inference keys; `y_*` targets only when add_parsing)."""

    def __init__(self, encodings, add_parsing=False, rhl_win_size=7):
        self._encodings = encodings
        self._add_parsing = add_parsing
        self._rhl_win_size = rhl_win_size
        self._case_cache = {}
        self._wid_cache = {}

    def _char_codes(self, ch):
        """(char id, case id) of one character; case: 1 symbol, 2 upper, 3 lower (utils.py:294-304)."""
        hit = self._case_cache.get(ch)
        if hit is None:
            lo = ch.lower()
            case = 1 if lo == ch.upper() else (2 if lo != ch else 3)
            hit = (self._encodings.char2int.get(lo, self._encodings.char2int["<UNK>"]), case)
            self._case_cache[ch] = hit
        return hit

    def _char_tables(self, cps):
        """code point -> (char id, case id) lookup arrays, grown on demand (the per-character rule of
        `_char_codes` is evaluated once per distinct character, then applied by fancy indexing)."""
        top = int(cps.max()) + 1 if len(cps) else 1
        tab = getattr(self, "_cp_tab", None)
        if tab is None or tab.shape[1] < top:
            new = np.full((2, max(top, 256)), -1, dtype=np.int64)
            if tab is not None:
                new[:, :tab.shape[1]] = tab
            self._cp_tab = tab = new
        missing = np.unique(cps[tab[0, cps] < 0])
        for cp in missing.tolist():
            tab[:, cp] = self._char_codes(chr(cp))
        return tab

    def cp_table32(self, missing=None):
        """int32 [2, top] (char id, case id) by code point, -1 = not seen yet, for the C featuriser
        (csrc/host_collate.c); `missing` = code points to add (the per-character rule of `_char_codes`)."""
        tab = getattr(self, "_cp_tab32", None)
        if tab is None:
            tab = self._cp_tab32 = np.full((2, 256), -1, dtype=np.int32)
            missing = list(range(32, 127)) + list(missing or [])
        if missing:
            top = max(missing) + 1
            if top > tab.shape[1]:
                new = np.full((2, max(top, 2 * tab.shape[1])), -1, dtype=np.int32)
                new[:, :tab.shape[1]] = tab
                tab = new
            for cp in set(missing):
                tab[:, cp] = self._char_codes(chr(cp))
            self._cp_tab32 = tab  # a new array object each time it grows: readers keep a consistent one
        return tab

    def collate_fn(self, batch, targets=True):
        """`targets=False` (used by `process`, which drops them anyway) skips the y_* tensors."""
        enc = self._encodings
        sent_len = np.fromiter((len(s.words) for s in batch), dtype=np.int64, count=len(batch))
        words = [w for s in batch for w in s.words]
        texts = [w.word for w in words]
        word_len = np.fromiter(map(len, texts), dtype=np.int64, count=len(words))
        B, L, N, W = len(batch), int(sent_len.max()), len(words), int(word_len.max())
        K = len(words[0].emb)
        w2i, unk = enc.word2int, enc.word2int["<UNK>"]
        # flat character stream (UTF-32 code points of all words) -> (word, position) scatter
        cps = np.frombuffer("".join(texts).encode("utf-32-le", "surrogatepass"), dtype=np.uint32).astype(np.int64)
        tab = self._char_tables(cps)
        x_word = np.zeros((N, W), dtype=np.int64)
        x_word_case = np.zeros((N, W), dtype=np.int64)
        inside = np.arange(W)[None, :] < word_len[:, None]   # position < word length
        x_word_masks = inside.astype(np.float64)
        hm = host.load()
        if hm is not None:
            hm.fill_ragged(np.ascontiguousarray(tab[0, cps]), word_len, x_word, W)
            hm.fill_ragged(np.ascontiguousarray(tab[1, cps]), word_len, x_word_case, W)
        else:  # row-major order of `inside` = the order of the character stream
            x_word[inside] = tab[0, cps]
            x_word_case[inside] = tab[1, cps]
        sb = np.repeat(np.arange(B), sent_len)
        si = np.arange(N) - np.repeat(np.cumsum(sent_len) - sent_len, sent_len)
        x_sent = np.zeros((B, L), dtype=np.int64)
        x_sent_masks = np.zeros((B, L), dtype=np.float64)
        # word ids (lower-cased look-up, utils.py:284-289) through a cache keyed by the raw token
        wid = self._wid_cache
        ids = list(map(wid.get, texts))
        if None in ids:
            if len(wid) > (1 << 20):
                wid.clear()
            for i, t in enumerate(texts):
                if ids[i] is None:
                    ids[i] = wid[t] = w2i.get(t.lower(), unk)
        x_sent[sb, si] = ids
        x_sent_masks[sb, si] = 1
        x_lang_sent = np.fromiter((s.lang_id + 1 for s in batch), dtype=np.int64, count=B)
        x_lang_word = np.repeat(x_lang_sent, sent_len)
        embs = []
        for k in range(K):
            rows = [w.emb[k] for w in words]
            e = None
            r0 = rows[0]
            if hm is not None and isinstance(r0, np.ndarray) and r0.ndim == 1:
                e = np.empty((N, r0.shape[0]), dtype=r0.dtype)
                try:
                    hm.stack_rows(rows, e)  # one memcpy per word
                except (ValueError, TypeError, BufferError):  # rows of other sizes / kinds: the general way below
                    e = None
            if e is None:
                try:
                    e = np.concatenate(rows).reshape(N, -1)
                except ValueError:  # not arrays of one length: let NumPy sort it out the slow way
                    e = np.asarray(rows)
            embs.append(torch.from_numpy(np.ascontiguousarray(e)))
        out = {
            "x_sent": torch.from_numpy(x_sent), "x_lang_sent": torch.from_numpy(x_lang_sent),
            "x_word": torch.from_numpy(x_word), "x_word_case": torch.from_numpy(x_word_case),
            "x_lang_word": torch.from_numpy(x_lang_word), "x_sent_len": torch.from_numpy(sent_len),
            "x_word_len": torch.from_numpy(word_len), "x_sent_masks": torch.from_numpy(x_sent_masks),
            "x_word_masks": torch.from_numpy(x_word_masks), "x_word_embeddings": embs,
        }
        if not targets:
            return out

        def targets(attr, table):
            y = np.zeros((B, L), dtype=np.int64)
            y[sb, si] = np.fromiter((table.get(getattr(w, attr), 0) for w in words), dtype=np.int64, count=N)
            return torch.from_numpy(y)

        out["y_upos"] = targets("upos", enc.upos2int)
        out["y_xpos"] = targets("xpos", enc.xpos2int)
        out["y_attrs"] = targets("attrs", enc.attrs2int)
        if self._add_parsing:
            heads = np.fromiter((int(w.head) for w in words), dtype=np.int64, count=N)
            y_head = np.zeros((B, L), dtype=np.int64)
            y_head[sb, si] = heads
            y_rhl = np.zeros((B, L), dtype=np.int64)
            y_rhl[sb, si] = np.clip(heads - si + self._rhl_win_size, 0, self._rhl_win_size * 2 - 1)
            out["y_head"] = torch.from_numpy(y_head)
            out["y_label"] = targets("label", enc.label2int)
            out["y_rhl"] = torch.from_numpy(y_rhl)
        return out


def unpack(data, sizes, max_size, device=None):
    """reference: utils.py:17-32 — packed [N, D] -> padded [B, max_size, D], zeros at pads."""
    sizes = np.asarray(sizes, dtype=np.int64)
    out = torch.zeros((len(sizes), max_size, data.shape[-1]), dtype=torch.float32, device=data.device)
    b = torch.from_numpy(np.repeat(np.arange(len(sizes)), sizes)).to(data.device)
    i = torch.from_numpy(np.arange(int(sizes.sum())) - np.repeat(np.cumsum(sizes) - sizes, sizes)).to(data.device)
    out[b, i] = data.to(torch.float32)
    return out


class ChuLiuEdmondsDecoder:
    """reference: utils.py:400-577.  decode(score[B, L, L+1] probabilities, lens[B]) -> list of
    int64 arrays; runs on the GPU (cube_mst_decode); raises like the reference's assertion
    (utils.py:556-562) when a sentence has no single-rooted tree of non-zero probability."""

    def __init__(self, device=None):
        self._device = device

    def decode_device(self, probs, lens):
        """probs f32 [B, L, L+1] and lens int32 [B] on the GPU -> (heads int32 [B, L], status int32 [B])."""
        lib = hip.load()
        B, L, T = probs.shape
        assert T == L + 1 and probs.dtype == torch.float32 and probs.is_contiguous()
        heads = torch.empty((B, L), dtype=torch.int32, device=probs.device)
        status = torch.empty((B,), dtype=torch.int32, device=probs.device)
        nbytes = lib.cube_mst_workspace_bytes(B, L)
        ws = torch.empty((nbytes + 15) // 16 * 2, dtype=torch.int64, device=probs.device)
        with torch.cuda.device(probs.device):  # launch on the stream of the device that holds the scores
            hip.check(lib.cube_mst_decode(probs.data_ptr(), lens.data_ptr(), B, L, heads.data_ptr(), status.data_ptr(),
                                          ws.data_ptr(), ws.numel() * 8, hip.stream_handle()), "cube_mst_decode")
        return heads, status

    def decode(self, score, lens):
        dev = self._device or ("cuda:%d" % torch.cuda.current_device())
        probs = torch.as_tensor(np.ascontiguousarray(score, dtype=np.float32)).to(dev)
        lens_d = torch.as_tensor(np.asarray(lens, dtype=np.int32)).to(dev)
        heads, status = self.decode_device(probs, lens_d)
        heads, status = heads.cpu().numpy(), status.cpu().numpy()
        if (status != 0).any():
            raise AssertionError("no single-rooted tree with non-zero probability for sentence(s) %s"
                                 % np.where(status != 0)[0].tolist())
        return [heads[b, :int(lens[b])].astype(np.int64) for b in range(len(lens))]
