"""External word-embedding providers that fill `word.emb` before the parser runs (reference: cube/networks/lm.py).

`LMHelperHF` mirrors the reference's transformer provider (lm.py:121-211): the tokenizer and the transformer are the
caller's (HuggingFace objects loaded from local files, or any objects with the same two calls -- no model can be
fetched here, so none is named by default); what this module owns is the part of `_compute_we` around the model
call: piece bookkeeping, the 512-piece windows, and the sub-word -> word pooling with the cut into per-layer vectors,
which runs on the device (csrc/pool.hip, `cube_subword_pool`) on the hidden states where the model left them.
`LMHelperFT` mirrors the fastText provider (lm.py:34-66) around a loaded fastText model; the Languasito provider
(lm.py:69-118) needs a package that is not here and stays the caller's; `LMHelperDummy` and `LMHelperPrecomputed` are the
model-free ones.
"""
import ctypes

import numpy as np


class LMHelperDummy:
    """reference: cube/networks/lm.py:214-233 — one 1-dimensional layer, value 1.0."""

    def __init__(self, device="cpu", model=None):
        pass

    def get_embedding_size(self):
        return [1]

    def apply(self, document):
        for sent in document.sentences:
            for w in sent.words:
                w.emb = [[1.0]]

    def apply_raw(self, batch):
        return [[[1.0] for _ in sent] for sent in batch]


class LMHelperFT:
    """reference: lm.py:34-66 -- one 300-wide fastText vector per word.  `fasttext_model`: a loaded fastText model (any object
    with `get_word_vector(str)`); without it the reference's file `~/.fasttext/cc.<model>.300.bin` is loaded when the
    `fasttext` package and the file are present -- nothing is downloaded (the reference fetches the model, lm.py:41-48;
    there is no network here).  The look-up itself is the fastText library's: host code, not part of the device path."""

    def __init__(self, device="cpu", model=None, fasttext_model=None):
        if fasttext_model is None:
            import os
            filename = os.path.join(os.path.expanduser("~"), ".fasttext", "cc.%s.300.bin" % model)
            try:
                import fasttext
            except ImportError as e:
                raise NotImplementedError("LMHelperFT needs the fasttext package or a loaded model (fasttext_model=...)") from e
            if not os.path.exists(filename):
                raise FileNotFoundError("%s is missing and cannot be downloaded here (lm.py:41-48 would fetch it)" % filename)
            fasttext_model = fasttext.load_model(filename)
        self._fasttext = fasttext_model

    def get_embedding_size(self):
        return [300]

    def apply(self, document):
        for sent in document.sentences:
            for w in sent.words:
                w.emb = [self._fasttext.get_word_vector(w.word)]

    def apply_raw(self, batch):
        return [[self._fasttext.get_word_vector(w) for w in sent] for sent in batch]


class LMHelperPrecomputed:
    """The caller has already set `word.emb` (list of K float vectors of the given sizes)."""

    def __init__(self, sizes):
        self._sizes = list(sizes)

    def get_embedding_size(self):
        return self._sizes

    def apply(self, document):
        for sent in document.sentences:
            for w in sent.words:
                if w.emb is None or len(w.emb) != len(self._sizes):
                    raise ValueError("word.emb must hold %d vectors before the parser runs" % len(self._sizes))


class LMHelperHF:
    """reference: LMHelperHF (lm.py:121-211).  `device` must be a GPU: the pooling has no CPU path.

    model: a name / local directory for `transformers.AutoTokenizer` / `AutoModel` as in the reference (lm.py:122-129;
    offline only a local directory can work), or omit it and pass the two objects: `splitter(word)['input_ids']` =
    [start] + pieces + [end] (lm.py:155), `encoder(ids, return_dict=True)['hidden_states']` = K tensors
    [sentences, pieces, E] (lm.py:178-179)."""
    START, PAD, END = 0, 1, 2   # lm.py:147-149

    def __init__(self, device="cuda:0", model=None, splitter=None, encoder=None):
        import torch
        from cube_amd import hip
        self._device = torch.device(device)
        if self._device.type != "cuda":
            raise hip.CubeHipError("cube_amd LMHelperHF pools on an MI355X: give it device='cuda:N' (there is no CPU fallback)")
        hip.load()
        if splitter is None or encoder is None:
            from transformers import AutoModel, AutoTokenizer  # lm.py:122-129
            name = "xlm-roberta-base" if model is None else model
            splitter = AutoTokenizer.from_pretrained(name)
            encoder = AutoModel.from_pretrained(name, output_hidden_states=True)
        self._splitter = splitter
        self._xlmr = encoder
        self._xlmr.eval()
        self._xlmr.to(self._device)
        with torch.no_grad():
            tmp = self._xlmr(torch.tensor([[100]], device=self._device), return_dict=True)  # lm.py:133-135
        hs = tmp["hidden_states"]
        self._emb_size = [int(hs[0].shape[-1]) for _ in range(len(hs))]

    def get_embedding_size(self):
        return self._emb_size

    def _pieces(self, batch):
        """lm.py:145-168 -> (input ids int64 [S, max_len], per word: sentence, first piece position, number of pieces)."""
        sents, spans = [], []
        for ii, sent in enumerate(batch):
            c_sent, pos = [self.START], 1
            for w in sent.words:
                pieces = self._splitter(w.word)["input_ids"][1:-1]
                spans.append((ii, pos, len(pieces)))
                c_sent.extend(pieces)
                pos += len(pieces)
            c_sent.append(self.END)
            sents.append(c_sent)
        max_len = max(len(s) for s in sents)
        ids = np.full((len(sents), max_len), self.PAD, dtype=np.int64)
        for ii, s in enumerate(sents):
            ids[ii, :len(s)] = s
        return ids, np.asarray(spans, dtype=np.int32).reshape(-1, 3)

    def _compute_we_device(self, batch):
        """lm.py:141-197 with the pooling on the device: -> f32 device tensor [K, words of the batch, E]."""
        import torch
        from cube_amd import hip
        lib = hip.load()
        ids, spans = self._pieces(batch)
        K, E = len(self._emb_size), self._emb_size[0]
        n_words = spans.shape[0]
        with torch.cuda.device(self._device), torch.no_grad():
            x = torch.from_numpy(ids).to(self._device)
            parts = []
            for a in range(0, x.shape[1], 512):  # lm.py:171-181: the model sees 512 pieces at a time
                out = self._xlmr(x[:, a:a + 512], return_dict=True)
                parts.append([h.float() for h in out["hidden_states"]])
            layers = [p.contiguous() for p in parts[0]] if len(parts) == 1 else \
                [torch.cat([p[k] for p in parts], dim=1).contiguous() for k in range(K)]
            out = torch.empty((K, n_words, E), dtype=torch.float32, device=self._device)
            if n_words:
                sp = torch.from_numpy(np.ascontiguousarray(spans.T)).to(self._device)   # [3, n_words]: sentence, position, count
                ptrs = (ctypes.c_void_p * K)(*[t.data_ptr() for t in layers])
                S, P = layers[0].shape[0], layers[0].shape[1]
                hip.check(lib.cube_subword_pool(ptrs, K, P * E, E, E, sp[0].data_ptr(), sp[1].data_ptr(), sp[2].data_ptr(),
                                                n_words, out.data_ptr(), hip.stream_handle()), "cube_subword_pool")
                self._keep = (layers, sp)  # alive until the next call: the launch is asynchronous
        return out

    def _compute_we(self, batch):
        """reference signature (lm.py:141): one float32 vector of K*E values per word of the batch."""
        we = self._compute_we_device(batch).permute(1, 0, 2).contiguous().cpu().numpy()   # [words, K, E]
        return [we[i].reshape(-1) for i in range(we.shape[0])]

    def apply(self, doc):
        """lm.py:199-208: per sentence (a batch of one, as the reference calls the model), `word.emb` = K vectors."""
        for sent in doc.sentences:
            if not sent.words:
                continue
            we = self._compute_we_device([sent]).permute(1, 0, 2).contiguous().cpu().numpy()   # [words, K, E]
            for ii, w in enumerate(sent.words):
                w.emb = [we[ii, kk] for kk in range(we.shape[1])]

    def apply_raw(self, batch):
        pass
