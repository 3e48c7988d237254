"""MI355X-native joint tagger + dependency parser (inference).

Host-side mirror of the reference `Parser` (cube/networks/parser.py:22-95, 109-233,
400-448): same constructor, `load`, `to`, `forward`, `_get_labels` and `process` signatures
and results, but every tensor operation of the path runs in hand-written gfx950 kernels
behind the C ABI of include/cube_hip.h.

Precision (`Parser.precision`): "bf16x3" (default) keeps every activation and weight as a
hi/lo pair of bf16 planes and forms each product as hi*hi + hi*lo + lo*hi on the matrix cores
with fp32 accumulation — fp32-grade results, checked directly against the reference's own
outputs (tests/golden) to the north-star tolerance; "bf16" is the single-pass mode (one
plane, 3x fewer MFMAs) whose arithmetic contract is oracle/parser_oracle.py quant="bf16".
In bf16x3 mode the character-CNN layers >= 1 (inputs = gate outputs, |x| < 1) run in two-slice
int8 fixed point (`char_int8`, default on; cube_gemm_i8x2 in include/cube_hip.h): exact int32
accumulation of the slice products, 2.6e-5 on the reference's arc scores.

Per batch: 1 upload of int32 indices + bf16 external embeddings, ~45 asynchronous kernel
launches on the current HIP stream, 1 download of five int32 arrays — the reference's three
mid-batch syncs (parser.py:125, 425, 428-431) are gone because unpack, soft-max, MST decoding
and the head gather for label scoring all stay on the device.
"""
import ctypes
import os

import numpy as np
import torch

from cube_amd import hip, host
from cube_amd.networks import packing
from cube_amd.networks.packing import p32, p128
from cube_amd.networks.utils import MorphoCollate, MorphoDataset, ChuLiuEdmondsDecoder  # noqa: F401


_STAGE_STREAMS = {}   # (device type, index) -> the three stage streams shared by all models on that device (_init_pipeline)


class _TimedLaunch:
    """HIP events around one launch on the current stream (`_HipNet.start_kernel_timing`)."""

    def __init__(self, timing):
        self.timing = timing

    def __enter__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def __exit__(self, *exc):
        self.b.record()
        self.timing["pairs"].append((self.a, self.b))


class _NoTiming:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_TIMING = _NoTiming()


class _Buffers:
    """Grow-only, zero-initialised flat device buffers with fixed leading dimensions, so the
    padding columns no kernel writes stay zero for the life of the model.  bf16 buffers are
    `hip.Pair`s (hi plane + lo plane when the precision is bf16x3)."""

    def __init__(self, device, split):
        self.device = device
        self.split = split
        self._b = {}
        self.streams = ()  # side streams whose kernels may still use a buffer when it is replaced

    def get(self, name, rows, ld, dtype=torch.bfloat16):
        need = max(rows * ld, 1)
        t = self._b.get(name)
        numel = 0 if t is None else (t.hi.numel() if isinstance(t, hip.Pair) else t.numel())
        if t is None or numel < need:
            if t is not None:
                # a larger batch replaces the buffer while kernels of the previous batch may still read
                # the old one on another stream: tell the allocator, so the memory is not handed out
                # again before that work is done
                for plane in ((t.hi, t.lo) if isinstance(t, hip.Pair) else (t,)):
                    for s in self.streams:
                        if plane is not None and plane.is_cuda:
                            plane.record_stream(s)
            if dtype == torch.bfloat16:
                t = hip.Pair(torch.zeros(need, dtype=dtype, device=self.device),
                             torch.zeros(need, dtype=dtype, device=self.device) if self.split else None)
            else:
                t = torch.zeros(need, dtype=dtype, device=self.device)
            self._b[name] = t
        return t


class _HipNet:
    """What Parser and Tagger share: weight packing / upload, batch preparation, the character-level
    and word-level gated CNN trunk.  `_root` = 1 when every sentence gets a learned root row in
    front (parser), 0 otherwise (tagger)."""
    _root = 1
    _pack = staticmethod(packing.pack_parser)

    def __init__(self, config, encodings, language_codes=None, ext_word_emb=0):
        self._config = config
        self._char_layers = getattr(config, "char_layers", 3)  # WordGram's default (modules.py:426)
        self._encodings = encodings
        if not isinstance(ext_word_emb, (list, tuple)):
            ext_word_emb = [ext_word_emb]
        self._ext_word_emb = list(ext_word_emb)
        self._num_langs = encodings.num_langs
        self._language_codes = language_codes
        self._decoder = ChuLiuEdmondsDecoder()
        self._device = torch.device("cpu")
        self._pk_cpu = None
        self._pk = None
        self._lay = None
        self._buf = None
        self.training = False
        self.precision = "bf16x3"  # or "bf16"; set before .to(device)
        # character-CNN layers >= 1 in int8x2 fixed point (cube_gemm_i8x2; bf16x3 mode only): their inputs
        # are gate outputs, |x| < 1.  False = bf16x3 there too; CUBE_CHAR_I8=0 does the same.
        self.char_int8 = os.environ.get("CUBE_CHAR_I8", "1") != "0"
        # BiLSTM recurrence with the hidden-state weights resident in VGPRs + LDS and h . W_hh in two-slice int8 fixed
        # point (cube_bilstm_recurrent_i8; bf16x3 mode only).  False / CUBE_LSTM_I8=0: bf16x3, W_hh streamed from L2.
        self.lstm_int8 = os.environ.get("CUBE_LSTM_I8", "1") != "0"
        # word-level gated convolutions of layers >= 1 in the same int8x2 fixed point as the character CNN (their inputs
        # are gate outputs; the residual stream stays fp32).  False / CUBE_WORD_I8=0: bf16x3.
        self.word_int8 = os.environ.get("CUBE_WORD_I8", "1") != "0"
        # ... and layer 0, whose input is the concatenation [word | char | lang | ext], after one pass that turns it into
        # 16-bit fixed point with a scale per column block (cube_quant_i8x2).  Needs word_int8.  OFF by default: +1.7 %
        # sentences/s at config 1 (187.5k -> 190.8k in an A/B on one box), but embeddings in fixed point over their full
        # range are ~10x noisier than bf16x3 (UPOS logits 4e-5 -> 9e-5 relative, arc scores 2.7e-5 -> 3.6e-5) and config 0
        # then differs from the reference's text in 21 of 1000 sentences where the parity suite allows 20 (14 lines with
        # the default): decisions come first.  CUBE_WORD0_I8=1 / word0_int8 = True: on.
        self.word0_int8 = os.environ.get("CUBE_WORD0_I8", "0") != "0"
        self._timing = None
        self._streams = None
        self._layouts = {}

    # ---- reference-compatible module surface ----------------------------------------------
    def eval(self):
        self.training = False
        return self

    def load_state_dict(self, state_dict, strict=True):
        enc = self._encodings
        sizes = dict(upos=len(enc.upos2int), xpos=len(enc.xpos2int), attrs=len(enc.attrs2int),
                     labels=len(enc.label2int))
        self._sizes = sizes
        self._pk_cpu, self._lay = self._pack(state_dict, self._config, sizes, self._ext_word_emb)
        self._pk = None
        if self._device.type == "cuda":
            self.to(self._device)

    def load(self, model_path, device="cpu"):
        """reference: parser.py:400-402 (Lightning checkpoint, only ['state_dict'] is read)."""
        ckpt = torch.load(model_path, map_location="cpu", weights_only=True)
        self.load_state_dict(ckpt["state_dict"])
        self.to(device)

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            # the reference defaults to 'cpu'; this implementation has no CPU path
            self._device = device
            return self
        lib = hip.load()
        self._device = device
        with torch.cuda.device(device):
            hip.check(lib.cube_init(), "cube_init")  # per-device set-up of the library (include/cube_hip.h)
        if self._pk_cpu is not None:
            if self.precision not in ("bf16x3", "bf16"):
                raise ValueError("precision must be 'bf16x3' or 'bf16'")
            self._split = self.precision == "bf16x3"
            self._pk = packing.to_device(self._pk_cpu, device, with_lo=self._split)
            self._buf = _Buffers(device, self._split)
            self._layouts = {}
            self._buf.streams = self._streams or ()
        return self

    def save_packed(self, path):
        """The packed weights of this model (after `load` / `load_state_dict`) as a file that a host in another language
        reads through cube_packfile_* (include/cube_hip.h) instead of running the Python packer.  -> number of entries."""
        if self._pk_cpu is None:
            raise ValueError("load weights first (load / load_state_dict)")
        return packing.save_packed(self._pk_cpu, path, self._lay)

    def _require_gpu(self):
        if self._pk is None or self._device.type != "cuda":
            raise hip.CubeHipError("cube_amd %s runs only on an MI355X: load weights and call .to('cuda:N') "
                                   "(there is no CPU fallback)" % type(self).__name__)

    # ---- live kernel timing (bench.py roofline): HIP events on the launch stream ----------
    def start_kernel_timing(self, tag, every=1):
        """HIP events around the launches tagged `tag` from now on; every > 1: around every `every`-th of them only (an
        event pair costs the stream ~2.5 us -- the launch behind it waits for the kernel before it to END, where back-to-back
        launches of one stream otherwise overlap their ramp-down and ramp-up: with all four character-CNN launches of a
        step timed, a 20-step region ran 1.5 % slower than the same region untimed)."""
        self._timing = dict(tag=tag, pairs=[], every=max(1, int(every)), seen=0)

    @hip.on_device
    def stop_kernel_timing(self):
        t, self._timing = self._timing, None
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in t["pairs"]]
        return dict(tag=t["tag"], n=len(ms), ms_mean=float(np.mean(ms)) if ms else 0.0,
                    ms_min=float(np.min(ms)) if ms else 0.0)

    def _timed(self, tag):
        t = self._timing
        if t is None or t["tag"] != tag:
            return _NO_TIMING
        t["seen"] += 1
        if (t["seen"] - 1) % t["every"]:
            return _NO_TIMING
        return _TimedLaunch(t)

    # ---- batch preparation -----------------------------------------------------------------
    _PART_NAMES = ("char_idx", "case_idx", "lang_word", "word_len", "rowmap_word", "word_idx", "lang_rows", "root_rows",
                   "lens", "word_rows", "dup_dst", "dup_src")  # (the last two: C featuriser with `dedup_words` only)

    def _prepare_host(self, X):
        """reference-format batch dict (MorphoCollate.collate_fn) -> the host half of a prepared batch:
        one packed int32 array (the ten index arrays of `_PART_NAMES`, each padded to 4 elements), the
        external embeddings scattered to their padded [B*L, Epad] fp32 rows, the ragged character-CNN
        groups.  `_featurize` builds the same arrays straight from the Sentence objects in C."""
        x_sent = X["x_sent"].cpu().numpy()
        sl = X["x_sent_len"].cpu().numpy().astype(np.int64)
        x_word = X["x_word"].cpu().numpy()
        B, L = x_sent.shape
        N, W = x_word.shape
        root = self._root
        T = L + root
        lang_sent = X["x_lang_sent"].cpu().numpy()
        sb = np.repeat(np.arange(B), sl)
        si = np.arange(N) - np.repeat(np.cumsum(sl) - sl, sl)
        word_idx = np.zeros((B, T), dtype=np.int32)
        word_idx[:, root:] = x_sent
        ch = self._char_layout(X["x_word_len"].cpu().numpy().astype(np.int64), W)
        order = ch["order"]
        x_case = X["x_word_case"].cpu().numpy()
        char_idx = np.concatenate([x_word[order[s:e], :q].reshape(-1) for s, e, q in ch["gather"]])
        case_idx = np.concatenate([x_case[order[s:e], :q].reshape(-1) for s, e, q in ch["gather"]])
        l0 = self._pk_cpu.get("char_l0")
        if l0 is not None:  # layer 0 by table look-up: one (character, case) combination id per position
            char_idx, case_idx = char_idx * l0["n_case"] + case_idx, np.zeros(0, dtype=np.int64)
        parts = [
            char_idx, case_idx, X["x_lang_word"].cpu().numpy()[order], X["x_word_len"].cpu().numpy()[order],
            (sb * T + si + root)[order], word_idx.reshape(-1), np.repeat(lang_sent, T), np.arange(B) * T, sl,
            ch["word_rows"].reshape(-1),
        ]
        sizes = [int(p.size) for p in parts]
        offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in sizes])])
        packed = np.zeros(int(offs[-1]), dtype=np.int32)
        for p, o, s in zip(parts, offs[:-1], sizes):
            packed[o:o + s] = p
        embs = []
        for k, e in enumerate(X["x_word_embeddings"]):
            ek = self._pk_cpu["ext"][k]
            pad = np.zeros((B * L, ek["Epad"]), dtype=np.float32)
            pad[sb * L + si, :ek["E"]] = e.cpu().numpy().astype(np.float32, copy=False)
            embs.append(pad)
        return dict(packed=packed, sizes=sizes, B=B, L=L, T=T, N=N, W=W, embs=embs, conv_groups=ch["groups"],
                    char_rows=ch["rows"], y_upos=X.get("y_upos"), staging=None)

    def _featurize(self, batch, collate):
        """list of Sentence -> the host half of a prepared batch (as `_prepare_host`), in one C pass over
        the objects (csrc/host_collate.c `featurize`: the counterpart of MorphoCollate.collate_fn,
        utils.py:232-330, fused with the index packing), embeddings written into pinned staging memory.
        -> None when the C module is missing or the batch is not in the fast format (word.emb not a list of
        contiguous float32 vectors): the caller then goes through collate_fn + `_prepare_host`."""
        hm = host.load()
        if hm is None or not isinstance(collate, MorphoCollate) or not isinstance(batch, list) or not batch:
            return None
        B = len(batch)
        L = hm.max_words(batch) if hasattr(hm, "max_words") else max(len(s.words) for s in batch)
        if L < 1:
            return None
        pk = self._pk_cpu
        l0 = pk.get("char_l0")
        dims = tuple((ek["E"], ek["Epad"]) for ek in pk["ext"])
        st = self._staging_acquire([B * L * ep for _, ep in dims], 64 * B * L)
        n = self._char_layers
        groups = (hip.ConvGroups * n)()
        enc = collate._encodings
        tab = collate.cp_table32()
        # raw token -> word id: a C-side table of the helper module (one per collate = per vocabulary), else the dict
        wid_cache = getattr(collate, "_wid_table", None)
        if wid_cache is None:
            wid_cache = collate._wid_table = hm.wid_table_new() if hasattr(hm, "wid_table_new") else collate._wid_cache
        try:
            for _ in range(4):
                try:
                    r = hm.featurize(batch, enc.word2int, wid_cache, enc.word2int["<UNK>"], tab, self._root, n,
                                     0 if l0 is None else l0["n_case"], L, tuple(b.numpy() for b in st["emb"]), dims,
                                     ctypes.addressof(groups), self._host_threads, st["ints"].numpy(),
                                     1 if self.dedup_words else 0)
                except TypeError:
                    self._staging_release(st)
                    return None
                if isinstance(r, tuple):
                    break
                if isinstance(r, list):
                    tab = collate.cp_table32(r)  # code points seen for the first time: extend the table, run again
                else:  # the index buffer was too small (very long words): r = the int32 elements it must hold
                    st["ints"] = torch.empty(int(r) + int(r) // 4, dtype=torch.int32, pin_memory=self._device.type == "cuda")
            else:
                raise RuntimeError("featurize: the character table does not settle")
        except BaseException:
            # ValueError (a word that is not a str, a sentence that changed under us), MemoryError, ...: the set goes back
            # to the ring, or six such errors would leave no staging set for any later call
            if st["busy"]:
                self._staging_release(st)
            raise
        n_ints, sizes, N, W, rows = r
        packed = st["ints"].numpy()[:n_ints]
        embs = [b.numpy()[:B * L * ep].reshape(B * L, ep) for b, (_, ep) in zip(st["emb"], dims)]
        return dict(packed=packed, sizes=list(sizes), B=B, L=L, T=L + self._root, N=N, W=W, embs=embs,
                    conv_groups=[groups[l] for l in range(n)], _groups_array=groups, char_rows=list(rows), y_upos=None,
                    staging=st)

    _host_threads = 8
    # Words of a batch with the same spelling and language go through the character CNN once; the other occurrences
    # copy the first one's char-level vector (bit-identical: a row's result does not depend on the other rows).  Only
    # the C featuriser builds the lists (`_prepare_host`, the collate_fn route, keeps one row per word).
    dedup_words = True
    _N_STAGING = 6

    def _staging_acquire(self, emb_sizes, n_ints=0):
        """A set of pinned host buffers for one batch's external embeddings and packed indices, out of a ring; a set is reused
        only after the H->D copies that read it have completed (its event)."""
        import threading
        ring = self.__dict__.setdefault("_staging", [])
        lock = self.__dict__.setdefault("_staging_lock", threading.Lock())
        with lock:
            # a free set whose uploads have completed; else a new set while the ring may grow; else, in rotation, the free
            # set whose uploads were enqueued longest ago (the first free one would be the set just released, whose event --
            # the upload of the PREVIOUS batch -- is the one wait that is sure to block: 2.3 ms per batch, measured)
            k0 = self.__dict__.get("_staging_next", 0)
            order = [ring[(k0 + j) % len(ring)] for j in range(len(ring))]
            free = [c for c in order if not c["busy"]]
            st = next((c for c in free if c["event"] is None or c["event"].query()), None)
            if st is None and len(ring) < self._N_STAGING:
                st = dict(emb=[], ints=torch.empty(0, dtype=torch.int32), busy=False, event=None)
                ring.append(st)
            if st is None:
                if not free:
                    raise hip.CubeHipError("more than %d featurised batches are waiting for their upload" % self._N_STAGING)
                st = free[0]
            self._staging_next = (next(k for k, c in enumerate(ring) if c is st) + 1) % len(ring)
            st["busy"] = True
        if st["event"] is not None:
            st["event"].synchronize()
            st["event"] = None
        while len(st["emb"]) < len(emb_sizes):
            st["emb"].append(torch.empty(0, dtype=torch.float32))
        pin = self._device.type == "cuda"
        for k, need in enumerate(emb_sizes):
            if st["emb"][k].numel() < need:
                st["emb"][k] = torch.empty(need, dtype=torch.float32, pin_memory=pin)
        if st["ints"].numel() < n_ints:
            st["ints"] = torch.empty(n_ints, dtype=torch.int32, pin_memory=pin)
        return st

    @staticmethod
    def _staging_release(st, event=None):
        st["event"] = event
        st["busy"] = False

    @hip.on_device
    def _upload(self, h):
        """host half of a prepared batch -> device batch: one int32 upload, one fp32 upload per LM layer
        (from pinned staging memory when `_featurize` made it), hi/lo split on the device (cube_split_f32)."""
        self._require_gpu()
        dev = self._device
        lib = hip.load()
        try:
            ints = torch.from_numpy(h["packed"]).to(dev, non_blocking=True)
            offs = np.concatenate([[0], np.cumsum([(s + 3) // 4 * 4 for s in h["sizes"]])])
            bt = {n: ints[o:o + s] for n, o, s in zip(self._PART_NAMES, offs[:-1], h["sizes"])}
            embs = []
            # LM layers of one width land in ONE buffer [K, B*L, Epad] (constant stride: their MLPs' first layers are one
            # batched launch, `_trunk`)
            sizes = {e.size for e in h["embs"]}
            whole = None
            if len(h["embs"]) > 1 and len(sizes) == 1:
                n1 = sizes.pop()
                whole = hip.Pair(torch.empty(len(h["embs"]) * n1, dtype=torch.bfloat16, device=dev),
                                 torch.empty(len(h["embs"]) * n1, dtype=torch.bfloat16, device=dev) if self._split else None)
            for k, e in enumerate(h["embs"]):
                src = torch.from_numpy(e).reshape(-1)
                x = src.to(dev, non_blocking=True)
                if whole is not None:
                    pair = hip.Pair(whole.hi[k * n1:(k + 1) * n1], None if whole.lo is None else whole.lo[k * n1:(k + 1) * n1])
                else:
                    pair = hip.Pair(torch.empty(x.numel(), dtype=torch.bfloat16, device=dev),
                                    torch.empty(x.numel(), dtype=torch.bfloat16, device=dev) if self._split else None)
                hip.check(lib.cube_split_f32(x.data_ptr(), x.numel(), pair.hi.data_ptr(), hip.ptr(pair.lo), hip.stream_handle()),
                          "cube_split_f32")
                embs.append(pair)
        except BaseException:
            if h.get("staging") is not None:  # a failed upload must not keep the pinned set out of the ring
                try:
                    torch.cuda.synchronize(dev)  # (copies already enqueued may still read it)
                except Exception:
                    pass
                self._staging_release(h["staging"])
            raise
        if h["staging"] is not None:
            ev = torch.cuda.Event()
            ev.record()
            self._staging_release(h["staging"], ev)
        bt.update(B=h["B"], L=h["L"], T=h["T"], N=h["N"], W=h["W"], embs=embs, embs_whole=whole, conv_groups=h["conv_groups"],
                  char_rows=h["char_rows"], _keep=h.get("_groups_array"),
                  n_dup=int(h["sizes"][10]) if len(h["sizes"]) > 10 else 0)
        if h.get("y_upos") is not None:
            bt["y_upos"] = h["y_upos"].to(dev).to(torch.int32).contiguous()
        return bt

    def _prepare(self, X):
        """reference-format batch dict (MorphoCollate.collate_fn) -> device index arrays.
        One packed int32 upload + one fp32 upload per LM layer."""
        self._require_gpu()
        return self._upload(self._prepare_host(X))

    def _host_batch(self, batch, collate):
        """list of Sentence -> host half of the prepared batch: the C featuriser, or the reference-format
        collate followed by `_prepare_host`."""
        h = self._featurize(batch, collate)
        if h is None:
            X = collate.collate_fn(batch, targets=False) if isinstance(collate, MorphoCollate) else collate.collate_fn(batch)
            X.pop("y_upos", None)
            h = self._prepare_host(X)
        return h

    def _char_layout(self, word_len, W):
        """Ragged layouts of the character CNN.  The masked mean (modules.py:479-483) reads positions
        < len of every gated layer and each Conv1d(k=5) looks two positions ahead, so layer l (of n)
        only has to produce positions < len + 2*(n-1-l) of a word, from positions < len + 2*(n-l) of
        its input — the padded tail of short words is never computed.  Words are grouped by length
        (<= 32 groups); layout k (k = 0: embeddings, k = l+1: output of layer l) keeps
        min(W, len_g + 2*(n-k)) rows per word of group g, groups one after the other.
        -> order (sorted word -> original word), per-layer cube_conv_groups, rows per layout,
        word_rows [n, N] (first row of sorted word i in the output of layer l), gather list."""
        n = self._char_layers
        key = np.minimum(np.maximum(word_len, 1), hip.MAX_CONV_GROUPS - 1)
        order = np.argsort(key, kind="stable")
        ks, starts, counts = np.unique(key[order], return_index=True, return_counts=True)
        gmax = np.array([max(int(word_len[order[s:s + c]].max()), 1) for s, c in zip(starts, counts)])
        seq = np.minimum(W, gmax[None, :] + 2 * (n - np.arange(n + 1))[:, None])       # [n+1 layouts, G]
        rows_g = seq * counts[None, :]
        row0 = np.concatenate([np.zeros((n + 1, 1), dtype=np.int64), np.cumsum(rows_g, axis=1)[:, :-1]], axis=1)
        groups = []
        for l in range(n):
            cg = hip.ConvGroups()
            cg.n = len(ks)
            for g in range(len(ks)):
                cg.row0_in[g], cg.n_rows[g], cg.seq_in[g] = int(row0[l, g]), int(rows_g[l, g]), int(seq[l, g])
                cg.row0_out[g], cg.seq_out[g], cg.word0[g] = int(row0[l + 1, g]), int(seq[l + 1, g]), int(starts[g])
            groups.append(cg)
        j = np.arange(len(word_len)) - np.repeat(starts, counts)                          # index inside the group
        word_rows = np.stack([np.repeat(row0[l + 1], counts) + j * np.repeat(seq[l + 1], counts) for l in range(n)])
        gather = [(int(s), int(s + c), int(q)) for s, c, q in zip(starts, counts, seq[0])]
        return dict(order=order, groups=groups, rows=[int(r) for r in rows_g.sum(axis=1)],
                    word_rows=word_rows.astype(np.int32), gather=gather)

    # ---- the device pipeline ---------------------------------------------------------------
    def _gather(self, dst, ld, col0, width, table, idx, rows, n):
        """embedding lookup / row copy into a concat buffer, every plane (hi[, lo])."""
        lo = dst.lo is not None and table.lo is not None
        hip.check(hip.load().cube_gather_rows(dst.hi.data_ptr(), hip.ptr(dst.lo) if lo else None, ld, col0, width,
                                              table.hi.data_ptr(), hip.ptr(table.lo) if lo else None, table.hi.shape[1],
                                              hip.ptr(idx), hip.ptr(rows), n, hip.stream_handle()), "cube_gather_rows")

    def _gather_batch(self, items):
        """several `_gather`s that do not write the same elements as ONE launch (cube_gather_rows_batch):
        items = (dst Pair, ld, col0, width, table Pair, idx, rows, n)."""
        descs = (hip.GatherDesc * len(items))()
        for d, (dst, ld, col0, width, table, idx, rows, n) in zip(descs, items):
            lo = dst.lo is not None and table.lo is not None
            d.dst, d.dst_lo, d.ld_dst, d.col0, d.width = dst.hi.data_ptr(), (dst.lo.data_ptr() if lo else None), ld, col0, width
            d.table, d.table_lo, d.ld_table = table.hi.data_ptr(), (table.lo.data_ptr() if lo else None), table.hi.shape[1]
            d.idx, d.dst_rows, d.n = hip.ptr(idx), hip.ptr(rows), n
        hip.check(hip.load().cube_gather_rows_batch(descs, len(items), hip.stream_handle()), "cube_gather_rows_batch")

    @staticmethod
    def _zero(pair, numel):
        pair.hi[:numel].zero_()
        if pair.lo is not None:
            pair.lo[:numel].zero_()

    def _char_level(self, bt):
        """Character CNN of WordGram (modules.py:450-481) up to the masked mean, in ragged per-length
        layouts (`_char_layout`).  -> (pool Pair [N, pool_w], pool_w): the (x + res) masked mean per word."""
        lib, pk, lay, buf = hip.load(), self._pk, self._lay, self._buf
        st = hip.stream_handle()
        gather = self._gather
        lo = hip.ptr
        N, W = bt["N"], bt["W"]
        ch = lay.char_half
        rows = bt["char_rows"]
        l0 = pk.get("char_l0")
        nchar = len(pk["char_convs"])
        # int8x2 planes for the gate outputs when every GEMM layer reads one (layer 0 = table look-ups)
        i8 = (self.char_int8 and self._split and l0 is not None and lay.xc_w % 64 == 0 and ch % 16 == 0 and
              all("W8" in cv for cv in pk["char_convs"][1:]))
        if l0 is None:
            xc0 = buf.get("xc0", rows[0], lay.xc0_w)
            gather(xc0, lay.xc0_w, 0, lay.char_emb, pk["tok_emb"], bt["char_idx"], None, rows[0])
            gather(xc0, lay.xc0_w, lay.char_emb, lay.case_emb, pk["case_emb"], bt["case_idx"], None, rows[0])
        # every gated layer keeps its own output g_l: the next layer reads it, and the masked mean
        # sums them (no fp32 residual stream through HBM at N*W rows).  The language channels are
        # not materialised: they enter through the per-(language, tap mask) bias table.
        if i8:  # int8 [rows, xc_w] per plane = the bytes of a bf16 buffer of half the width
            gl = [buf.get("xc_q%d" % l, rows[l + 1], lay.xc_w // 2) for l in range(nchar)]
        else:
            gl = [buf.get("xc_g%d" % l, rows[l + 1], lay.xc_w) for l in range(nchar)]
        src, src_ld = (xc0, lay.xc0_w) if l0 is None else (None, 0)
        for l, cv in enumerate(pk["char_convs"]):
            if l == 0 and l0 is not None:
                # embeddings + first gated layer as table look-ups (cube_char_conv0_lookup): HBM/L2-bound
                # instead of a K = 5*288 GEMM over every character position
                with self._timed("char_conv0"):
                    fn = lib.cube_char_conv0_lookup_i8 if i8 else lib.cube_char_conv0_lookup
                    hip.check(fn(l0["T"].data_ptr(), l0["btab"].data_ptr(), bt["char_idx"].data_ptr(),
                                 bt["lang_word"].data_ptr(), ctypes.pointer(bt["conv_groups"][0]),
                                 l0["taps"], l0["half"], l0["n_combo"], l0.get("zero_code", -1), gl[0].hi.data_ptr(),
                                 lo(gl[0].lo), lay.xc_w, st), "cube_char_conv0_lookup")
                src, src_ld = gl[0], lay.xc_w
                continue
            if i8:
                with self._timed("char_conv"):
                    hip.gemm(src, src_ld, rows[l], cv["Kc"], cv["W8"], cv["N"], cv["bias"], hip.EPI_GATE, out0=gl[l],
                             ldo0=lay.xc_w, col0=0, taps=cv["taps"], seq_len=W, n_valid=cv["half"],
                             bias_tab=cv["bias_tab"], row_group=bt["lang_word"], groups=bt["conv_groups"][l], stream=st,
                             w_scale=cv["w_scale"])
                src, src_ld = gl[l], lay.xc_w
                continue
            with self._timed("char_conv" if l > 0 else "char_conv0"):
                hip.gemm(src, src_ld, rows[l], cv["Kc"], cv["W"], cv["N"], cv["bias"], hip.EPI_GATE, out0=gl[l],
                         ldo0=lay.xc_w, col0=0, taps=cv["taps"], seq_len=W, n_valid=cv["half"],
                         bias_tab=cv["bias_tab"], row_group=bt["lang_word"], groups=bt["conv_groups"][l], stream=st)
            src, src_ld = gl[l], lay.xc_w
        pool_w = p32(ch)
        pool = buf.get("pool", N, pool_w)
        hip.masked_mean(gl, [1.0] * (nchar - 1) + [2.0], lay.xc_w, bt["word_len"], N, W, ch, pool, pool_w,
                        word_rows=bt["word_rows"], stream=st, i8=i8)
        return pool, pool_w

    def _trunk(self, bt, final, final_w, lang_dsts, tap=None, tap_w=0, ext_b=None, ext_b_w=0, ext_b_col=0,
               want_aux=False):
        """Character CNN (WordGram, modules.py:450-484), external-embedding MLPs, word-level input
        assembly and the word-level gated CNN (parser.py:130-176 / tagger.py:102-148).
        final: Pair [B*T, final_w] that receives x+res of the last layer in columns [0, word_half);
        tap: Pair that receives x+res after layer `aux_softmax_location` (parser only);
        lang_dsts: (Pair, ld) buffers whose columns [word_half, word_half+lang) get the language
        embedding; ext_b: optional second destination of the projected external embeddings."""
        self._require_gpu()
        lib, pk, lay, cfg, buf = hip.load(), self._pk, self._lay, self._config, self._buf
        st = hip.stream_handle()
        gather = self._gather
        lo = hip.ptr  # lo-plane pointer helper: hip.ptr(pair.lo) is None in single-pass mode
        B, L, T, N, W = bt["B"], bt["L"], bt["T"], bt["N"], bt["W"]
        BT, BL = B * T, B * L
        ch, wh, lg, P = lay.char_half, lay.word_half, lay.lang, lay.P

        pool, pool_w = self._char_level(bt)
        # ---- word level input XW0 = [word | char | lang | ext] (+ root row = _r_emb for the parser)
        xw0 = buf.get("xw0", BT, lay.xw0_w)
        if N + bt.get("n_dup", 0) != BL:
            # pad positions keep a zero char-level vector (unpack, utils.py:17-32); every other element
            # of XW0 is rewritten below for every row.  Without pads nothing needs clearing.
            self._zero(xw0, BT * lay.xw0_w)
        # the look-ups that depend on the batch's index arrays only, in one launch: word embeddings and the language
        # embedding into XW0, the language embedding into the buffers the later layers read it from
        self._gather_batch([(xw0, lay.xw0_w, 0, lay.word_emb, pk["word_emb"], bt["word_idx"], None, BT),
                            (xw0, lay.xw0_w, lay.xw0_lang, lg, pk["lang_emb"], bt["lang_rows"], None, BT)] +
                           [(dstb, ldb, wh, lg, pk["lang_emb"], bt["lang_rows"], None, BT) for dstb, ldb in lang_dsts])
        po = pk["char_pre_out"]
        hip.gemm(pool, pool_w, N, po["K"], po["W"], po["N"], po["bias"], hip.EPI_BF16_TANH, out0=xw0, ldo0=lay.xw0_w,
                 col0=lay.xw0_char, n_valid=ch, rowmap=bt["rowmap_word"], stream=st)
        n_dup = bt.get("n_dup", 0)
        if n_dup:  # repeated words: the char-level vector of the first occurrence, row to row inside XW0
            src_hi, src_lo = hip._planes(xw0, lay.xw0_char)
            hip.check(lib.cube_gather_rows(xw0.hi.data_ptr(), lo(xw0.lo), lay.xw0_w, lay.xw0_char, ch, src_hi,
                                           src_lo if xw0.lo is not None else None, lay.xw0_w, bt["dup_src"].data_ptr(),
                                           bt["dup_dst"].data_ptr(), n_dup, st), "cube_gather_rows")
        # ---- external embeddings: sum_k MLP_k(e_k) / K, tanh   (parser.py:130-138, tagger.py:111-119)
        cat = pk.get("ext_h1cat") if bt.get("embs_whole") is not None else None
        n_ext, hn = len(pk["ext"]), pk["ext"][0]["h1"]["N"]
        ext_h = buf.get("ext_h", BL, hn * (n_ext if cat else 1))
        ld_part = 8 * ((P + 7) // 8)
        parts = n_ext if cat else 1
        ld_acc = ld_part * parts
        ext_acc = buf.get("ext_acc", BL, ld_acc, torch.float32)
        if cat:  # the K first layers as one batched launch, side by side in ext_h; the K second layers as another,
            # side by side in ext_acc; cube_ext_finish sums them in the order of the LM layers
            ep, c2 = pk["ext"][0]["Epad"], pk["ext_h2cat"]
            hip.gemm(bt["embs_whole"], ep, BL, cat["K"], cat["W"], cat["N"], cat["bias"], hip.EPI_BF16_TANH, out0=ext_h,
                     ldo0=hn * n_ext, n_valid=hn, stream=st, batch=n_ext, stride_a=BL * ep, stride_w=cat["N"] * cat["K"],
                     stride_bias=cat["N"], stride_col0=hn)
            hip.gemm(ext_h, hn * n_ext, BL, c2["K"], c2["W"], c2["N"], c2["bias"], hip.EPI_F32, out0=ext_acc, ldo0=ld_acc,
                     n_valid=P, stream=st, batch=n_ext, stride_a=hn, stride_w=c2["N"] * c2["K"], stride_bias=c2["N"],
                     stride_col0=ld_part)
        else:
            for k, ek in enumerate(pk["ext"]):
                h1, h2 = ek["h1"], ek["h2"]
                hip.gemm(bt["embs"][k], ek["Epad"], BL, h1["K"], h1["W"], h1["N"], h1["bias"], hip.EPI_BF16_TANH,
                         out0=ext_h, ldo0=h1["N"], n_valid=h1["N"], stream=st)
                hip.gemm(ext_h, h1["N"], BL, h2["K"], h2["W"], h2["N"], h2["bias"], hip.EPI_F32, out0=ext_acc, ldo0=ld_acc,
                         n_valid=P, accumulate=1 if k > 0 else 0, stream=st)
        hip.check(lib.cube_ext_finish(ext_acc.data_ptr(), ld_acc, B, L, self._root, P, len(pk["ext"]), parts, ld_part,
                                      xw0.hi.data_ptr(), lo(xw0.lo), lay.xw0_w, lay.xw0_ext,
                                      None if ext_b is None else ext_b.hi.data_ptr(),
                                      None if ext_b is None else lo(ext_b.lo), ext_b_w, ext_b_col, st), "cube_ext_finish")
        if self._root:
            gather(xw0, lay.xw0_w, 0, lay.xw0_w, pk["r_emb"], None, bt["root_rows"], B)
        out = {}
        if want_aux:
            ax = pk["aux"]
            aux = buf.get("aux", BT, ax["n_valid"], torch.float32)[:BT * ax["n_valid"]].view(BT, ax["n_valid"])
            hip.gemm(xw0, lay.xw0_w, BT, ax["K"], ax["W"], ax["N"], ax["bias"], hip.EPI_F32, out0=aux,
                     ldo0=ax["n_valid"], n_valid=ax["n_valid"], a_col=lay.xw0_char, stream=st)
            out["aux"] = aux
        # ---- word-level gated CNN
        nword = len(pk["word_convs"])
        # int8x2 planes between the layers (layer 0, whose inputs are embeddings, stays bf16x3 and writes them)
        w8 = (self.word_int8 and self._split and nword > 1 and lay.xw_w % 64 == 0 and wh % 16 == 0 and
              all("W8" in cv and cv["Kc"] == lay.xw_w for cv in pk["word_convs"][1:]))
        if w8:  # int8 [BT, xw_w] per plane = the bytes of a bf16 buffer of half the width
            xwa, xwb = buf.get("xwq_a", BT, lay.xw_w // 2), buf.get("xwq_b", BT, lay.xw_w // 2)
        else:
            xwa, xwb = buf.get("xwa", BT, lay.xw_w), buf.get("xwb", BT, lay.xw_w)
        resw = buf.get("resw", BT, wh, torch.float32)
        src, src_ld = xw0, lay.xw0_w
        # layer 0 in int8x2 as well: XW0 (embeddings, tanh outputs, the root row) -> 16-bit fixed point with one scale per
        # column block (packing.xw0_block_scales, folded into the layer's fixed-point weights)
        cv0 = pk["word_convs"][0]
        w8_0 = w8 and self.word0_int8 and "W8" in cv0 and "in_inv_scale" in cv0 and cv0["Kc"] == lay.xw0_w
        if w8_0:   # rows of Kc8 bytes per plane (64-channel chunks); the columns past xw0_w are never written and stay zero
            k8 = cv0["Kc8"]
            xw0q = buf.get("xw0q", BT, k8 // 2)
            hip.check(lib.cube_quant_i8x2(xw0.hi.data_ptr(), lo(xw0.lo), lay.xw0_w, BT, lay.xw0_w, cv0["in_inv_scale"].data_ptr(),
                                          xw0q.hi.data_ptr(), xw0q.lo.data_ptr(), k8, st), "cube_quant_i8x2")
            src, src_ld = xw0q, k8
        for l, cv in enumerate(pk["word_convs"]):
            last = l == nword - 1
            dst = None if last else (xwa if l % 2 == 0 else xwb)
            aux_here = tap is not None and (l + 1 == cfg.aux_softmax_location)
            o1, o2 = (tap if aux_here else None), (final if last else None)
            if w8 and (l > 0 or w8_0):
                hip.gemm(src, src_ld, BT, cv["Kc8"], cv["W8"], cv["N"], cv["bias"], hip.EPI_GATE, out0=dst, ldo0=lay.xw_w,
                         col0=0, taps=cv["taps"], seq_len=T, n_valid=cv["half"], res=resw, ld_res=wh, res_mode=0 if l == 0 else 1,
                         out1=o1, ldo1=tap_w, col1=0, out2=o2, ldo2=final_w, col2=0, bias_tab=cv.get("bias_tab"),
                         row_group=bt["lang_rows"] if "bias_tab" in cv else None, stream=st, w_scale=cv["w_scale"])
                src, src_ld = dst, lay.xw_w
                continue
            hip.gemm(src, src_ld, BT, cv["Kc"], cv["W"], cv["N"], cv["bias"], hip.EPI_GATE, out0=dst, ldo0=lay.xw_w,
                     col0=0, taps=cv["taps"], seq_len=T, n_valid=cv["half"], res=resw, ld_res=wh,
                     res_mode=0 if l == 0 else 1, out1=o1, ldo1=tap_w, col1=0, out2=o2, ldo2=final_w, col2=0,
                     bias_tab=cv.get("bias_tab"), row_group=bt["lang_rows"] if "bias_tab" in cv else None, stream=st,
                     out0_int8=w8 and dst is not None)
            src, src_ld = dst, lay.xw_w
        return out

    def _tag_heads(self, bt, slot, x, x_w, upos_key, emb_col, row_off=0, cap_rows=None):
        """UPOS logits -> arg-max (or gold tags) -> UPOS embedding into columns [emb_col, emb_col+64)
        of `x` -> ATTRS | XPOS logits; arg-max of all three (parser.py:180-192, tagger.py:153-161).
        row_off / cap_rows: the slot's result buffers hold `cap_rows` positions and this batch writes its rows from
        `row_off` on (several batches whose later stages run together, `Parser.submit`)."""
        lib, pk, buf, st = hip.load(), self._pk, self._buf, hip.stream_handle()
        B, T = bt["B"], bt["T"]
        BT, root = B * T, self._root
        cap = BT if cap_rows is None else cap_rows
        sfx = "_%d" % slot
        n_u, n_x, n_a = self._sizes["upos"], self._sizes["xpos"], self._sizes["attrs"]
        upos_log = buf.get("upos_log" + sfx, cap, n_u, torch.float32)[row_off * n_u:(row_off + BT) * n_u].view(BT, n_u)
        q = pk[upos_key]
        hip.gemm(x, x_w, BT, q["K"], q["W"], q["N"], q["bias"], hip.EPI_F32, out0=upos_log, ldo0=n_u, n_valid=n_u,
                 stream=st)
        tag_all = buf.get("tag_idx" + sfx, 3, cap, torch.int32)[:3 * cap].view(3, cap)  # upos, attrs, xpos
        tag_idx = tag_all[:, row_off:row_off + BT]
        ue = pk["upos_emb"]
        if "y_upos" in bt:  # gold tags given (training-time behaviour of forward, parser.py:185-188)
            hip.check(lib.cube_argmax_rows(upos_log.data_ptr(), n_u, 0, n_u, BT, tag_idx[0].data_ptr(), st), "cube_argmax_rows")
            emb_idx = torch.zeros((B, T), dtype=torch.int32, device=self._device)
            emb_idx[:, root:] = bt["y_upos"]
            emb_idx = emb_idx.view(-1)
            bt["_gold_idx"] = emb_idx  # lives as long as the batch (the gather below is asynchronous)
            self._gather(x, x_w, emb_col, 64, ue, emb_idx, None, BT)
        else:  # arg-max and the look-up of its embedding in one launch
            lo_ok = x.lo is not None and ue.lo is not None
            hip.check(lib.cube_tag_argmax(upos_log.data_ptr(), n_u, BT, 0, n_u, tag_idx[0].data_ptr(), 0, 0, None,
                                          ue.hi.data_ptr(), ue.lo.data_ptr() if lo_ok else None, ue.hi.shape[1], 64,
                                          x.hi.data_ptr(), x.lo.data_ptr() if lo_ok else None, x_w, emb_col, st), "cube_tag_argmax")
        q = pk["attrs_xpos"]
        n_ax = q["n_valid"]
        ax_log = buf.get("ax_log" + sfx, cap, n_ax, torch.float32)[row_off * n_ax:(row_off + BT) * n_ax].view(BT, n_ax)
        hip.gemm(x, x_w, BT, q["K"], q["W"], q["N"], q["bias"], hip.EPI_F32, out0=ax_log, ldo0=n_ax, n_valid=n_ax,
                 stream=st)
        hip.check(lib.cube_tag_argmax(ax_log.data_ptr(), n_ax, BT, 0, n_a, tag_idx[1].data_ptr(), n_a, n_x, tag_idx[2].data_ptr(),
                                      None, None, 0, 0, None, None, 0, 0, st), "cube_tag_argmax")
        return dict(tag_idx=tag_idx, tag_all=tag_all,
                    upos_logits=upos_log.view(B, T, n_u)[:, root:, :], ax_logits=ax_log.view(B, T, n_ax)[:, root:, :],
                    upos=tag_idx[0].view(B, T)[:, root:], attrs=tag_idx[1].view(B, T)[:, root:],
                    xpos=tag_idx[2].view(B, T)[:, root:])


class Parser(_HipNet):
    _root = 1
    _pack = staticmethod(packing.pack_parser)

    def __init__(self, config, encodings, language_codes=None, ext_word_emb=0):
        super().__init__(config, encodings, language_codes, ext_word_emb)
        self._decoder = ChuLiuEdmondsDecoder()
        self._streams = None

    def _front(self, bt, slot=0, want_aux=False, row_off=0, cap_rows=None):
        """GEMM-heavy first phase: char CNN, ext MLPs, word CNN, morphology heads.  Leaves the LSTM
        input in xl0_{slot} and the tag logits in slot buffers -- from row `row_off` of buffers that hold
        `cap_rows` positions when several batches share the slot's later stages (`submit`)."""
        pk, lay, buf, st = self._pk, self._lay, self._buf, hip.stream_handle()
        BT = bt["B"] * bt["T"]
        cap = BT if cap_rows is None else cap_rows
        wh, lg, P = lay.word_half, lay.lang, lay.P
        # XL0 = [x_parse | lang | ext] (LSTM input), XM = [hidden | lang] (morphology input)
        xl0_w = p32(wh + lg + P)
        xl0_all = buf.get("xl0_%d" % slot, cap, xl0_w)
        # XL0: x_parse, lang and ext columns are rewritten for every word row each batch; only the ext
        # columns of the root rows must read zero (parser.py:195-197), and they do as long as the
        # (rows, T) layout of this slot's buffer has not changed since it was last cleared (root rows sit at multiples of T)
        lay_key = "xl0_layout_%d" % slot
        if self._layouts.get(lay_key) != (cap, bt["T"]):
            self._zero(xl0_all, cap * xl0_w)
            self._layouts[lay_key] = (cap, bt["T"])
        o = row_off * xl0_w
        xl0 = xl0_all if row_off == 0 else hip.Pair(xl0_all.hi[o:], None if xl0_all.lo is None else xl0_all.lo[o:])
        xm_w = p32(wh + lg)
        xm = buf.get("xm", BT, xm_w)
        # PM = [pre_morpho | lang | upos_emb], the input of the morphology heads (parser.py:178-192)
        pm_w = p32(wh + lg + 64)
        pm = buf.get("pm", BT, pm_w)
        out = self._trunk(bt, xl0, xl0_w, [(xm, xm_w), (xl0, xl0_w), (pm, pm_w)], tap=xm, tap_w=xm_w, ext_b=xl0, ext_b_w=xl0_w,
                          ext_b_col=wh + lg, want_aux=want_aux)
        q = pk["pre_morpho"]
        hip.gemm(xm, xm_w, BT, q["K"], q["W"], q["N"], q["bias"], hip.EPI_BF16_TANH, out0=pm, ldo0=pm_w,
                 n_valid=q["n_valid"], stream=st)
        out.update(self._tag_heads(bt, slot, pm, pm_w, "upos", wh + lg, row_off, cap_rows))
        out.update(xl0=xl0_all, xl0_w=xl0_w)
        return out

    def _back(self, bt, fr, slot=0):
        """Latency-bound second phase: BiLSTM, pre_parsing, arc scorer + softmax.  Adds its outputs
        to `fr`."""
        lib, pk, lay, cfg, buf = hip.load(), self._pk, self._lay, self._config, self._buf
        st = hip.stream_handle()
        lo = hip.ptr
        B, L, T = bt["B"], bt["L"], bt["T"]
        BT = B * T
        lg, H = lay.lang, cfg.rnn_size
        # ---- BiLSTM (parser.py:194-202): per layer one projection GEMM + one persistent recurrent kernel
        xl_w = p32(2 * H + lg)
        xla, xlb = buf.get("xla", BT, xl_w), buf.get("xlb", BT, xl_w)
        self._gather_batch([(xla, xl_w, 2 * H, lg, pk["lang_emb"], bt["lang_rows"], None, BT),
                            (xlb, xl_w, 2 * H, lg, pk["lang_emb"], bt["lang_rows"], None, BT)])
        G2 = 2 * pk["lstm"][0]["G"]
        gx = buf.get("gx", BT, G2, torch.float32)
        src, src_ld = fr["xl0"], fr["xl0_w"]
        for l, ly in enumerate(pk["lstm"]):
            ih = ly["ih"]
            dst = xla if l % 2 == 0 else xlb
            hip.gemm(src, src_ld, BT, ih["K"], ih["W"], ih["N"], ih["bias"], hip.EPI_F32, out0=gx, ldo0=G2, n_valid=G2,
                     stream=st)
            if self.lstm_int8 and self._split and ly.get("whh8") is not None:
                # hidden-state weights resident in VGPRs + LDS, recurrent product in two-slice int8 fixed point
                w8 = ly["whh8"]
                hip.check(lib.cube_bilstm_recurrent_i8(gx.data_ptr(), w8["hi"].data_ptr(), w8["lo"].data_ptr(),
                                                       w8["scale"].data_ptr(), B, T, H, w8["lo_bits"], dst.hi.data_ptr(),
                                                       lo(dst.lo), xl_w, 0, st), "cube_bilstm_recurrent_i8")
            else:
                hip.check(lib.cube_bilstm_recurrent(gx.data_ptr(), ly["whh"].hi.data_ptr(), lo(ly["whh"].lo), B, T, H,
                                                    dst.hi.data_ptr(), lo(dst.lo), xl_w, 0, st), "cube_bilstm_recurrent")
            src, src_ld = dst, xl_w
        # ---- pre_parsing (parser.py:202-203) and the arc scorer (modules.py:518-533, 550-579)
        q = pk["pre_out"]
        pp_w = q["N"]
        pp = buf.get("pp_%d" % slot, BT, pp_w)  # per slot: read by the decode stage while the next batch is encoded
        hip.gemm(src, src_ld, BT, q["K"], q["W"], q["N"], q["bias"], hip.EPI_BF16_TANH, out0=pp, ldo0=pp_w,
                 n_valid=q["n_valid"], stream=st)
        # both biaffine input projections in one launch: HAL = [h1;1 | h2;1 | x1;1 | x2;1] (arc scorer
        # inputs, then the label scorer's; per slot: the label scorer reads it in the decode stage)
        q = pk["bi_in"]
        Dh, Dl = q["Dh"], q["Dl"]
        hal_w = 2 * Dh + 2 * Dl
        hal = buf.get("hal_%d" % slot, BT, hal_w)
        hip.gemm(pp, pp_w, BT, q["K"], q["W"], q["N"], q["bias"], hip.EPI_BF16_RELU, out0=hal, ldo0=hal_w,
                 n_valid=hal_w, stream=st)
        q = pk["head_bilin"]
        u = buf.get("u", BT, Dh)
        hip.gemm(hal, hal_w, BT, q["K"], q["W"], q["N"], None, hip.EPI_BF16, out0=u, ldo0=Dh, n_valid=Dh, stream=st)
        arc_logits = buf.get("arc_logits_%d" % slot, B * L, T, torch.float32)[:B * L * T].view(B, L, T)
        arc_probs = buf.get("arc_probs_%d" % slot, B * L, T, torch.float32)[:B * L * T].view(B, L, T)
        hip.check(lib.cube_biaffine_arc(u.hi.data_ptr(), lo(u.lo), Dh, hal.hi.data_ptr(), lo(hal.lo), hal_w, Dh, Dh, B, T,
                                        arc_logits.data_ptr(), arc_probs.data_ptr(), st), "cube_biaffine_arc")
        fr.update(hal=hal, hal_w=hal_w, hl_col=2 * Dh)
        fr.update(arc_logits=arc_logits, arc_probs=arc_probs, pp=pp, pp_w=pp_w)
        return fr

    def _encode(self, bt, want_aux=False, slot=0):
        return self._back(bt, self._front(bt, slot, want_aux), slot)

    def _label_scores(self, bt, pp, pp_w, heads_dev, labels_out=None, hl=None, hl_w=0, hl_col=0, slot=0):
        """parser.py:218-233: gather head rows + DeepBiaffine(pairwise=False), fused on device.
        hl (with its leading dimension and first column): the [x1;1 | x2;1] projections when the encode
        stage has already made them (`_back`); else they are computed here from `pp`."""
        lib, pk, st = hip.load(), self._pk, hip.stream_handle()
        B, L, T = bt["B"], bt["L"], bt["T"]
        BT = B * T
        q = pk["label_in"]
        Dl = q["D"]
        if hl is None:
            hl, hl_w, hl_col = self._buf.get("hl", BT, 2 * Dl), 2 * Dl, 0
            hip.gemm(pp, pp_w, BT, q["K"], q["W"], q["N"], q["bias"], hip.EPI_BF16_RELU, out0=hl, ldo0=2 * Dl,
                     n_valid=2 * Dl, stream=st)
        n_l = self._sizes["labels"]
        logits = self._buf.get("label_logits_%d" % slot, B * L, n_l, torch.float32)[:B * L * n_l].view(B, L, n_l)
        labels = labels_out if labels_out is not None else \
            self._buf.get("labels", B, L, torch.int32)[:B * L].view(B, L)
        lb = pk["label_bilin"]
        hl_hi, hl_lo = hip._planes(hl, hl_col)
        hip.check(lib.cube_label_score(hl_hi, hl_lo, hl_w, Dl, Dl, lb["W"].hi.data_ptr(),
                                       hip.ptr(lb["W"].lo), lb["bias"].data_ptr(), n_l, heads_dev.data_ptr(),
                                       bt["lens"].data_ptr(), B, L, logits.data_ptr(), labels.data_ptr(), st),
                  "cube_label_score")
        return logits, labels

    def _decode(self, bt, r, slot=0):
        """MST heads + labels for an encoded batch; packs [heads, labels, upos, xpos, attrs] as one
        int32 [5, B, L] device array (`r["packed"]`)."""
        B, L = bt["B"], bt["L"]
        packed = self._buf.get("packed_%d" % slot, 5, B * L, torch.int32)[:5 * B * L].view(5, B, L)
        status = self._buf.get("status_%d" % slot, 1, B, torch.int32)[:B]
        ws_bytes = hip.load().cube_mst_workspace_bytes(B, L)
        ws = self._buf.get("mst_ws_%d" % slot, 1, (ws_bytes + 7) // 8, torch.int64)
        hip.check(hip.load().cube_mst_decode(r["arc_probs"].data_ptr(), bt["lens"].data_ptr(), B, L,
                                             packed[0].data_ptr(), status.data_ptr(), ws.data_ptr(), ws.numel() * 8,
                                             hip.stream_handle()), "cube_mst_decode")
        label_logits, labels = self._label_scores(bt, r["pp"], r["pp_w"], packed[0], labels_out=packed[1],
                                                  hl=r["hal"], hl_w=r["hal_w"], hl_col=r["hl_col"], slot=slot)
        tag = r["tag_all"]   # (upos, attrs, xpos) of every position -> rows 2..4 = upos, xpos, attrs of the words
        hip.check(hip.load().cube_pack_tags(tag.data_ptr(), tag.shape[1], B, bt["T"], self._root, packed[2].data_ptr(),
                                            hip.stream_handle()), "cube_pack_tags")
        r.update(heads=packed[0], labels=labels, status=status, label_logits=label_logits, packed=packed, bt=bt)
        return r

    # ---- reference API ---------------------------------------------------------------------
    @hip.on_device
    def forward(self, X):
        """reference: parser.py:109-216.  Returns (arc logits [B,L,T], pre_parsing [B,T,pre],
        upos, xpos, attrs, aupos, axpos, aattrs) as float32 device tensors."""
        bt = self._prepare(X)
        self._last_bt = bt
        r = self._encode(bt, want_aux=True)
        B, L, T = bt["B"], bt["L"], bt["T"]
        n_u, n_x, n_a = self._sizes["upos"], self._sizes["xpos"], self._sizes["attrs"]
        n_pp = B * T * r["pp_w"]
        pre = hip.Pair(r["pp"].hi[:n_pp], None if r["pp"].lo is None else r["pp"].lo[:n_pp]).float()
        pre = pre.view(B, T, r["pp_w"])[:, :, :self._config.pre_parser_size]
        attrs, xpos = r["ax_logits"][:, :, :n_a], r["ax_logits"][:, :, n_a:n_a + n_x]
        aux = r["aux"].view(B, T, -1)[:, 1:, :]
        return (r["arc_logits"], pre, r["upos_logits"], xpos, attrs, aux[:, :, :n_u], aux[:, :, n_u:n_u + n_x],
                aux[:, :, n_u + n_x:n_u + n_x + n_a])

    @hip.on_device
    def _get_labels(self, labels, heads):
        """reference: parser.py:218-233.  `labels` = pre_parsing as returned by forward (the
        batch it came from must be the last one given to forward), `heads` = list of arrays."""
        bt = self._last_bt
        B, L, T = bt["B"], bt["L"], bt["T"]
        hp = np.zeros((B, L), dtype=np.int32)
        for b, h in enumerate(heads):
            hp[b, :len(h)] = np.asarray(h)
        pp_w = self._pk["pre_out"]["N"]
        pp = self._buf.get("pp_ext", B * T, pp_w)
        src = hip.Pair.split(labels.to(self._device).to(torch.float32), with_lo=self._split)
        pp.hi[:B * T * pp_w].view(B, T, pp_w)[:, :, :labels.shape[-1]] = src.hi
        if self._split:
            pp.lo[:B * T * pp_w].view(B, T, pp_w)[:, :, :labels.shape[-1]] = src.lo
        bt["_heads"] = torch.from_numpy(hp).to(self._device)  # kept with the batch: the kernel is asynchronous
        logits, _ = self._label_scores(bt, pp, pp_w, bt["_heads"])
        return logits

    @hip.on_device
    def decode_batch(self, X):
        """One batch of `process` on the device: everything up to the five int32 result arrays."""
        return self.decode_batch_prepared(self._prepare(X))

    @hip.on_device
    def decode_batch_prepared(self, bt):
        """Single-stream, in-order execution on the current stream (tests / forward-style use);
        returns device tensors (views into the workspace, valid until the next batch)."""
        return self._decode(bt, self._encode(bt))

    # ---- pipelined execution: GEMM-heavy phase of batch i+1 overlaps the latency-bound phase of i ----
    # Batches are taken in GROUPS of up to `PAIR`: each batch runs its own first stage (its own geometry: longest word,
    # its own ragged character layouts), writing the LSTM input and the tag decisions into the group's buffers behind
    # those of the batches before it; the latency-bound later stages -- input projections, the recurrent kernels,
    # pre_parsing, the arc scorer, MST decoding, the label scorer, the D->H copy -- then run ONCE for the group as one
    # batch of all its sentences.  Every row of those stages depends on its own sentence only (the bitwise
    # batch-independence the parity suite checks), so results do not change; what changes is that the 3 x T serial
    # recurrent steps, the decoder's task waves and a dozen launch-bound small GEMMs are paid once per group.
    PAIR = int(os.environ.get("CUBE_PAIR", "3"))  # batches per group (same sentence length; the later ones no larger than the first)
    # Grouping pays when the serial chain of stage 2 (3 layers x T recurrent steps of ~4.6 us) outlasts a first stage
    # (~1.1 ms for 8192 tokens): measured 25k -> 35k sentences/s at T = 129, nothing at T = 33 (the chip-time is the same,
    # and a group drains later at the end of a short run: 158k -> 152k in a 20-step region).  Below this length: no groups.
    PAIR_MIN_T = int(os.environ.get("CUBE_PAIR_MIN_T", "64"))
    N_SLOTS = 2 * PAIR + 1   # batches that may be in flight (submitted, not collected)

    def in_flight_for(self, T):
        """How many batches of sentence length T - 1 a caller should keep in flight: two groups and the batch being
        submitted.  Without groups that is 3 -- measured in a 20-step region: 3 in flight 158k sentences/s, 5 or 7 in
        flight 140..150k (the first stages run ahead and the last batches' later stages trail alone); in the steady
        state the depth makes no difference."""
        return 2 * (self.PAIR if T >= self.PAIR_MIN_T else 1) + 1
    N_GROUPS = N_SLOTS       # groups in flight: as many as batches, for when no two consecutive batches share a length

    def _init_pipeline(self):
        # Three streams, one per stage, and no more: how HIP maps streams to its (by default 4) hardware queues
        # depends on what else created streams in the process (RCCL under torch.distributed does), and two stages
        # that land on one queue run one after the other.  A decode stream per slot was measured: 117k..159k
        # sentences/s depending on GPU_MAX_HW_QUEUES and on whether RCCL was loaded; one decode stream:
        # 154k..159k in all four combinations, and config 2 no slower (16.7k).
        # ... and the SAME three for every model of the process on this device: streams created later (a second Parser,
        # a Tagger beside it) were seen to land two to a hardware queue, which runs their stages one after the other
        # (pipelined rate = one-stream rate, scripts/batch_size_sweep.py before this); sharing streams between models only
        # adds ordering between them.
        key = (self._device.type, self._device.index)
        if key not in _STAGE_STREAMS:
            _STAGE_STREAMS[key] = tuple(torch.cuda.Stream(device=self._device) for _ in range(3))
        self._streams = _STAGE_STREAMS[key]
        self._groups = [dict(free=None, host=None, pending=0, members=[], open=False, done=None) for _ in range(self.N_GROUPS)]
        self._open = None
        self._n_groups = 0
        self._buf.streams = self._streams

    @hip.on_device
    def submit(self, bt):
        """Enqueue one prepared batch on three HIP streams, one per stage: (1) char/word CNNs and tag
        heads (GEMMs that fill all CUs), (2) BiLSTM, pre_parsing and the arc scorer (a few persistent
        workgroups, serial steps), (3) MST decode, label scorer and the D->H copy of the results.  The
        stages of one batch run in order (events); the latency-bound stages 2 and 3 of earlier batches
        run under stage 1 of later ones, and they run once per GROUP of `PAIR` consecutive batches of the same
        sentence length (above): a batch's stages 2 and 3 are enqueued when its group is full, when a batch that does
        not fit the group arrives, or when `collect` asks for it.  Returns a ticket for `collect`.  At most `N_SLOTS`
        batches may be in flight (submitted, not collected)."""
        self._require_gpu()
        if self._streams is None:
            self._init_pipeline()
        B, L, T = bt["B"], bt["L"], bt["T"]
        if sum(x["pending"] for x in self._groups) >= self.N_SLOTS:
            raise hip.CubeHipError("Parser.submit: %d batches are in flight, collect() the oldest first" % self.N_SLOTS)
        g = self._open
        if g is not None:
            grp = self._groups[g]
            if grp["T"] != T or grp["rows"] + B * T > grp["cap"] or len(grp["members"]) >= grp["pair"]:
                self._close_group(g)
                g = None
        s1 = self._streams[0]
        cur = torch.cuda.current_stream()
        if g is None:
            # a group whose batches have all been collected (its pinned result buffer has been read): the one used
            # longest ago first; with fewer than N_SLOTS = N_GROUPS batches in flight there always is one
            free = [k for k in range(self.N_GROUPS) if not self._groups[k]["pending"]]
            g = min(free, key=lambda k: self._groups[k].get("serial", -1))
            grp = self._groups[g]
            self._n_groups += 1
            grp["serial"] = self._n_groups
            pair = self.PAIR if T >= self.PAIR_MIN_T else 1
            grp.update(members=[], open=True, T=T, L=L, rows=0, cap=pair * B * T, done=None, B=0, pair=pair)
            self._open = g
            if grp["free"] is not None:
                s1.wait_event(grp["free"])  # the later stages of the group that used these buffers last
        with torch.cuda.stream(s1):
            s1.wait_stream(cur)  # inputs uploaded on the caller's stream
            fr = self._front(bt, g, row_off=grp["rows"], cap_rows=grp["cap"])
            ev_front = torch.cuda.Event()
            ev_front.record(s1)
        # the ticket keeps the batch's device inputs alive until `collect`: they were allocated on the
        # caller's stream, and freeing them earlier would let the next upload reuse their memory
        # while the kernels of this batch (on the stage streams) still read it
        ticket = dict(group=g, serial=self._n_groups, b_off=grp["B"], B=B, L=L, bt=bt)
        grp["members"].append(ticket)
        grp.update(rows=grp["rows"] + B * T, B=grp["B"] + B, fr=fr, ev_front=ev_front, pending=grp["pending"] + 1)
        if len(grp["members"]) >= grp["pair"]:
            self._close_group(g)
        return ticket

    def _close_group(self, g):
        """Stages 2 and 3 for all the batches of group `g` as one batch."""
        grp = self._groups[g]
        if not grp["open"]:
            return
        grp["open"] = False
        if self._open == g:
            self._open = None
        s1, s2, s3 = self._streams
        B, L, T = grp["B"], grp["L"], grp["T"]
        members = grp["members"]
        with torch.cuda.stream(s2):
            s2.wait_event(grp["ev_front"])  # recorded after the group's last first stage; the stream is in order
            if len(members) == 1:
                vbt = members[0]["bt"]
            else:  # the index arrays of the group's rows, one after the other
                lang = self._buf.get("grp_lang_%d" % g, grp["cap"], 1, torch.int32)
                lens = self._buf.get("grp_lens_%d" % g, grp["cap"], 1, torch.int32)
                for m in members:
                    lang[m["b_off"] * T:(m["b_off"] + m["B"]) * T].copy_(m["bt"]["lang_rows"], non_blocking=True)
                    lens[m["b_off"]:m["b_off"] + m["B"]].copy_(m["bt"]["lens"], non_blocking=True)
                vbt = dict(B=B, L=L, T=T, lang_rows=lang[:B * T], lens=lens[:B])
            fr = dict(grp["fr"])
            tag = fr["tag_all"]
            root = self._root
            fr.update(upos=tag[0][:B * T].view(B, T)[:, root:], attrs=tag[1][:B * T].view(B, T)[:, root:],
                      xpos=tag[2][:B * T].view(B, T)[:, root:])
            r = self._back(vbt, fr, g)
            ev_back = torch.cuda.Event()
            ev_back.record(s2)
        with torch.cuda.stream(s3):
            s3.wait_event(ev_back)
            r = self._decode(vbt, r, g)
            n = 5 * B * L
            host = grp["host"]
            if host is None or host.numel() < n + B:
                host = torch.empty(n + B, dtype=torch.int32, pin_memory=True)
                grp["host"] = host
            # by a kernel's stores, not hipMemcpyAsync: a D->H copy waiting in the runtime's copy queue for this group's
            # kernels holds up the uploads of the next batches (cube_hip.h, cube_copy_to_pinned)
            lib = hip.load()
            hip.check(lib.cube_copy_to_pinned(r["packed"].data_ptr(), host.data_ptr(), 4 * n, hip.stream_handle()), "cube_copy_to_pinned")
            hip.check(lib.cube_copy_to_pinned(r["status"].data_ptr(), host.data_ptr() + 4 * n, 4 * B, hip.stream_handle()),
                      "cube_copy_to_pinned")
            done = torch.cuda.Event()
            done.record(s3)
        grp.update(done=done, free=done, vbt=vbt, fr=None)

    def collect(self, ticket):
        """Wait for a submitted batch: -> (int32 [5, B, L] = heads, labels, upos, xpos, attrs; status [B])."""
        g = ticket["group"]
        grp = self._groups[g]
        if not any(m is ticket for m in grp["members"]):
            raise hip.CubeHipError("Parser.collect: the ticket is not in flight")
        if grp["open"]:
            with torch.cuda.device(self._device):
                self._close_group(g)  # nobody else is coming: run the group's later stages now
        grp["done"].synchronize()
        ticket.pop("bt", None)
        Bg, L = grp["B"], grp["L"]
        host = grp["host"]
        n = 5 * Bg * L
        b0, b1 = ticket["b_off"], ticket["b_off"] + ticket["B"]
        out = host[:n].view(5, Bg, L)[:, b0:b1].numpy().copy(), host[n + b0:n + b1].numpy().copy()
        grp["members"] = [m for m in grp["members"] if m is not ticket]
        grp["pending"] -= 1  # at zero the group's next batches may overwrite the pinned buffer
        if not grp["pending"]:
            grp.pop("vbt", None)
        return out

    def _drain(self):
        """Forget batches that were submitted and never collected (an exception between submit and collect):
        wait for the stage streams, free every group."""
        if self._streams is not None:
            for st in self._streams:
                st.synchronize()
            for grp in self._groups:
                grp.update(pending=0, members=[], open=False, fr=None)
                grp.pop("vbt", None)
            self._open = None
        # staging sets that an exception left marked busy (between featurisation and upload): nothing reads them any more
        for st in self.__dict__.get("_staging", ()):
            if st["busy"]:
                if self._device.type == "cuda":
                    torch.cuda.synchronize(self._device)
                self._staging_release(st)

    def _write_back(self, batch, start, res, status):
        """parser.py:432-446: ids -> strings, into the Document."""
        if (status != 0).any():
            ids = np.asarray(start) + np.arange(len(status)) if np.ndim(start) == 0 else np.asarray(start)
            raise AssertionError("no single-rooted tree with non-zero probability for sentence(s) %s"
                                 % ids[status != 0].tolist())
        enc = self._encodings
        hm = host.load()
        if hm is not None and isinstance(batch, list):
            # one C call sets the five attributes of every word (csrc/host_collate.c `write_back`)
            hm.write_back(batch, np.ascontiguousarray(res, dtype=np.int32),
                          (enc.labels, enc.upos_list, enc.xpos_list, enc.attrs_list))
            return
        tabs = getattr(self, "_str_tabs", None)
        if tabs is None:  # id -> string tables as object arrays: one fancy-index per batch instead of a list lookup per word
            tabs = self._str_tabs = tuple(np.asarray(t, dtype=object) for t in
                                          (enc.labels, enc.upos_list, enc.xpos_list, enc.attrs_list))
        heads = res[0].astype(np.int64)
        lab, up, xp, at = tabs[0][res[1]], tabs[1][res[2]], tabs[2][res[3]], tabs[3][res[4]]
        for b, s in enumerate(batch):
            for w, h, l, u, x, a in zip(s.words, heads[b], lab[b], up[b], xp[b], at[b]):
                w.head, w.label, w.upos, w.xpos, w.attrs = h, l, u, x, a

    reports_results = True  # process / process_bucketed take on_result(batch, where, res[5, B, L])

    def _run_batches(self, batches, collate=None, on_result=None):
        """batches: iterable of (list of Sentence, position label for error messages).  Per batch the host
        featurises (one C pass over the objects, the embedding copies on worker threads), uploads, enqueues the kernels,
        then waits for the oldest batch in flight and writes its results into the Sentence objects;
        `N_SLOTS` batches are in flight on the device (a batch takes about two step times to pass the three
        stage streams, so two in flight would leave the first stage idle).
        `feature_thread` (off by default): featurise one batch ahead on a Python thread — measured slower: the
        two threads hand the GIL back and forth a hundred times per batch and each hand-over costs a wake-up."""
        from collections import deque
        collate = collate or self._collate
        self._drain()
        it = iter(batches)

        def feat(item):
            batch, start = item
            if not isinstance(batch, list):
                batch = list(batch)
            return batch, start, self._host_batch(batch, collate)

        pending = deque()

        def retire():
            b0, s0, t0 = pending.popleft()
            res, status = self.collect(t0)
            self._write_back(b0, s0, res, status)
            if on_result is not None:
                on_result(b0, s0, res)

        def enqueue(h):
            with self._upload_context():
                return self.submit(self._upload(h))

        if not self.feature_thread:
            for item in it:
                batch, start, h = feat(item)
                pending.append((batch, start, enqueue(h)))
                if len(pending) >= self.in_flight_for(h["T"]):
                    retire()
            while pending:
                retire()
            return
        from concurrent.futures import ThreadPoolExecutor
        pool = self.__dict__.get("_host_pool")
        if pool is None:
            pool = self._host_pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="cube-featurize")
        ahead = deque()
        for _ in range(2):
            item = next(it, None)
            if item is not None:
                ahead.append(pool.submit(feat, item))
        try:
            while ahead:
                batch, start, h = ahead.popleft().result()
                item = next(it, None)
                if item is not None:
                    ahead.append(pool.submit(feat, item))
                pending.append((batch, start, enqueue(h)))
                if len(pending) >= self.in_flight_for(h["T"]):
                    retire()
            while pending:
                retire()
        finally:
            for f in ahead:  # an exception above: let the worker finish, give its staging buffers back
                try:
                    r = f.result()
                    if r[2].get("staging") is not None:
                        self._staging_release(r[2]["staging"])
                except Exception:
                    pass

    feature_thread = os.environ.get("CUBE_FEATURE_THREAD", "0") == "1"
    upload_stream = os.environ.get("CUBE_UPLOAD_STREAM", "null")

    def _upload_context(self):
        """The stream the H->D copies of `process` go to."""
        import contextlib
        if self.upload_stream == "null" or self._device.type != "cuda":
            return contextlib.nullcontext()
        if self._streams is None:
            self._init_pipeline()
        if self.upload_stream == "front":
            return torch.cuda.stream(self._streams[0])
        up = self.__dict__.get("_up_stream")
        if up is None:
            up = self._up_stream = torch.cuda.Stream(device=self._device)
        return torch.cuda.stream(up)

    @hip.on_device
    def process_bucketed(self, doc, collate, max_tokens=8192, on_result=None):
        """Throughput mode for ragged documents: sentences sorted by length and
        packed into batches of at most `max_tokens` padded positions, results written back in place.
        NOT the reference's batch composition: scores differ from `process` in the way the
        reference's own scores differ between batch sizes (padding is visible to its convolutions
        and BiLSTM)."""
        from cube_amd.sharding import bucketed_batches
        self.eval()
        self._require_gpu()
        self._collate = collate
        sents = doc.sentences
        groups = bucketed_batches([len(s.words) for s in sents], max_tokens)
        self._run_batches((([sents[i] for i in idx], np.asarray(idx)) for idx in groups), on_result=on_result)
        return doc

    @hip.on_device
    def process(self, doc, collate, batch_size=4, num_workers=4, merge=True, max_tokens=32768, on_result=None):
        """reference: parser.py:404-448 — in-order batches of `batch_size` sentences; fills
        head/label/upos/xpos/attrs of every word of `doc` and returns it.  `in_flight_for(T)` device batches are
        in flight (3 for short sentences, 7 when later stages run per group of batches): featurisation and
        write-back of neighbours overlap the GPU.  `max_tokens` (default 32768 padded tokens per device batch) sets
        the footprint: 3.6 GB of device memory at the peak and ~100 MB of pinned staging at the default
        (INTEGRATION.md, section 3); 8192 takes a quarter of both at -5 % sentences/s.
        merge (no counterpart in the reference, results unchanged): reference batches with the same
        padding geometry (longest sentence, longest word) run together as one device batch of up to
        `max_tokens` padded tokens — every sentence still sees exactly the padding of its own
        reference batch, and sentences do not interact otherwise (`sharding.merged_reference_batches`)."""
        self.eval()
        self._require_gpu()
        self._collate = collate
        sents = doc.sentences
        starts = list(range(0, len(sents), batch_size))
        if not merge or len(starts) < 2:
            self._run_batches(((sents[start:start + batch_size], start) for start in starts), on_result=on_result)
            return doc
        from cube_amd.sharding import merged_reference_batches, sentence_batch_keys

        def cap(k):
            return min(1024, max_tokens // (k[0] + 1))
        groups = merged_reference_batches(sentence_batch_keys(sents, starts, batch_size, cap), batch_size, cap)

        def batches():
            for g in groups:
                ids = [i for r in g for i in range(starts[r], min(starts[r] + batch_size, len(sents)))]
                yield [sents[i] for i in ids], np.asarray(ids)
        self._run_batches(batches(), on_result=on_result)
        return doc
