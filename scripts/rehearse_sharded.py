"""Two (or more) ranks on ONE GPU, gloo for the gather: the sharded path with the real HIP parser against the same
document processed by one rank.  Launched from a shell (never from a process that has touched the GPU):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 scripts/rehearse_sharded.py"""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from tests import cases  # noqa: E402
from cube_amd import synth, sharding  # noqa: E402
from cube_amd.networks.parser import Parser  # noqa: E402
from cube_amd.networks.utils import MorphoCollate  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
cfg, enc, sd = cases.parser_model(cases.FULL)
p = Parser(cfg, enc, ext_word_emb=[300])
p.load_state_dict(sd)
p.to(dev)
col = MorphoCollate(enc)
rng = np.random.default_rng(5)
lens = np.clip(np.round(rng.normal(22, 10, size=203)), 1, 80).astype(np.int64)   # not a multiple of anything
doc = synth.make_document(enc, lens, seed=55, ext_word_emb=(300,))
for bucketed, bs in ((False, 4), (True, 4), (False, 32)):
    d = copy.deepcopy(doc)
    sharding.process_sharded(p, d, col, batch_size=bs, bucketed=bucketed)
    if rank == 0:
        ref = copy.deepcopy(doc)
        if bucketed:
            # the buckets of a shard are not the buckets of the whole document, and scores depend on the batch composition
            # (padding is visible to the reference's convolutions and BiLSTM) -- with random-init weights that moves a few
            # per cent of the heads: every word must carry a result; agreement with the single-rank run is printed
            p.process_bucketed(ref, col)
            assert all(w.label for s in d.sentences for w in s.words)
            same = np.mean([int(a.head) == int(b.head) for s, t in zip(d.sentences, ref.sentences) for a, b in zip(s.words, t.words)])
            print("bucketed over %d ranks: every word filled; heads equal to the single-rank bucketed run on %.1f %%" % (world, 100 * same))
        else:
            p.process(ref, col, batch_size=bs)
            for s, t in zip(d.sentences, ref.sentences):
                for a, b in zip(s.words, t.words):
                    assert (a.head, a.label, a.upos, a.xpos, a.attrs) == (b.head, b.label, b.upos, b.xpos, b.attrs)
            print("batch_size %d over %d ranks: identical to one rank (%d sentences, %d words)" % (bs, world, len(lens), int(lens.sum())))
dist.barrier()
dist.destroy_process_group()
