"""MI355X-native stand-alone tagger (UPOS / XPOS / ATTRS), inference.

Host-side mirror of the reference `Tagger` (cube/networks/tagger.py:22-67, 91-162, 288-329):
same constructor, `load`, `to`, `forward` and `process` signatures and results.  It is the
parser's first phase without the root row: the character CNN, the external-embedding MLPs and
the word-level gated CNN are the same kernels behind the same C ABI (include/cube_hip.h); only
the heads differ (UPOS reads [x+res | lang], XPOS / ATTRS read [upos_emb | lang | x+res]).
No CPU fallback: without libcubehip.so and an MI355X every call raises.
"""
import numpy as np
import torch

from cube_amd import hip
from cube_amd.networks import packing
from cube_amd.networks.packing import p32
from cube_amd.networks.parser import _HipNet
from cube_amd.networks.utils import MorphoCollate, MorphoDataset  # noqa: F401


class Tagger(_HipNet):
    _root = 0
    _pack = staticmethod(packing.pack_tagger)

    def _tag(self, bt, want_aux=False, slot=0):
        lay, buf = self._lay, self._buf
        BT = bt["B"] * bt["T"]
        wh, lg = lay.word_half, lay.lang
        # TU = [x+res | lang | upos_emb]
        tu_w = p32(wh + lg + 64)
        tu = buf.get("tu", BT, tu_w)
        out = self._trunk(bt, tu, tu_w, [(tu, tu_w)], want_aux=want_aux)
        out.update(self._tag_heads(bt, slot, tu, tu_w, "upos", wh + lg))
        return out

    # ---- reference API ---------------------------------------------------------------------
    @hip.on_device
    def forward(self, X):
        """reference: tagger.py:91-162.  Returns (upos, xpos, attrs, aupos, axpos, aattrs) logits
        [B, L, n] as float32 device tensors."""
        bt = self._prepare(X)
        self._last_bt = bt
        r = self._tag(bt, want_aux=True)
        B, L = bt["B"], bt["L"]
        n_u, n_x, n_a = self._sizes["upos"], self._sizes["xpos"], self._sizes["attrs"]
        attrs, xpos = r["ax_logits"][:, :, :n_a], r["ax_logits"][:, :, n_a:n_a + n_x]
        aux = r["aux"].view(B, L, -1)
        return (r["upos_logits"], xpos, attrs, aux[:, :, :n_u], aux[:, :, n_u:n_u + n_x],
                aux[:, :, n_u + n_x:n_u + n_x + n_a])

    @hip.on_device
    def tag_batch(self, X):
        """One batch of `process` on the device: -> int32 [3, B, L] = upos, xpos, attrs ids (host)."""
        bt = self._prepare(X)
        self._last_bt = bt  # device inputs stay alive while the asynchronous kernels read them
        r = self._tag(bt)
        B, L = bt["B"], bt["L"]
        packed = self._buf.get("packed_tags", 3, B * L, torch.int32)[:3 * B * L].view(3, B, L)
        packed[0].copy_(r["upos"])
        packed[1].copy_(r["xpos"])
        packed[2].copy_(r["attrs"])
        r["packed"] = packed
        return r

    @hip.on_device
    def process(self, doc, collate, upos=True, xpos=True, attrs=True, batch_size=32, num_workers=4, merge=True,
                max_tokens=8192):
        """reference: tagger.py:288-329 — in-order batches; fills upos / xpos / attrs of every word.
        The ids of a batch come back through pinned memory by an asynchronous copy, so the host
        prepares batch i+1 and writes batch i-1 back while the GPU works on batch i.
        merge: reference batches with the same padding geometry run together as one device batch, results
        unchanged (see Parser.process / sharding.merged_reference_batches)."""
        self.eval()
        if not (upos or xpos or attrs):
            raise Exception("To perform tagging at least one of 'upos', 'xpos' or 'attrs' must be set to True.")
        self._require_gpu()
        enc = self._encodings
        tabs = tuple(np.asarray(t, dtype=object) for t in (enc.upos_list, enc.xpos_list, enc.attrs_list))
        host = [None, None]

        def write_back(batch, h, done, B, L):
            done.synchronize()
            res = h[:3 * B * L].view(3, B, L).numpy()
            up, xp, at = tabs[0][res[0]], tabs[1][res[1]], tabs[2][res[2]]
            for b, s in enumerate(batch):
                for w, u, x, a in zip(s.words, up[b], xp[b], at[b]):
                    if upos:
                        w.upos = u
                    if xpos:
                        w.xpos = x
                    if attrs:
                        w.attrs = a

        sents = doc.sentences
        starts = list(range(0, len(sents), batch_size))
        if merge and len(starts) > 1:
            from cube_amd.sharding import merged_reference_batches, sentence_batch_keys

            def cap(k):
                return min(1024, max_tokens // max(k[0], 1))
            groups = merged_reference_batches(sentence_batch_keys(sents, starts, batch_size, cap), batch_size, cap)
            device_batches = [[s for r in g for s in sents[starts[r]:starts[r] + batch_size]] for g in groups]
        else:
            device_batches = [sents[a:a + batch_size] for a in starts]
        pending = None
        for k, batch in enumerate(device_batches):
            X = collate.collate_fn(batch, targets=False) if isinstance(collate, MorphoCollate) else collate.collate_fn(batch)
            X.pop("y_upos", None)
            packed = self.tag_batch(X)["packed"]
            _, B, L = packed.shape
            if host[k % 2] is None or host[k % 2].numel() < packed.numel():
                host[k % 2] = torch.empty(packed.numel(), dtype=torch.int32, pin_memory=True)
            host[k % 2][:packed.numel()].view(3, B, L).copy_(packed, non_blocking=True)
            done = torch.cuda.Event()
            done.record()
            if pending is not None:
                write_back(*pending)
            pending = (batch, host[k % 2], done, B, L)
        if pending is not None:
            write_back(*pending)
        return doc
