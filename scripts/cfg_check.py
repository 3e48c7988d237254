"""Race / equivalence screen for a change of GEMM tile configuration: the bench batches are decoded REPS
times in one process and every result must be bitwise equal to the first; the arrays are saved so that
runs of two builds (or tile selections) can be compared bitwise with scripts/npz_equal.py -- the
accumulation order per output element does not depend on the tile shape, so they must be identical.
    python scripts/cfg_check.py out.npz [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "nlp-cube_amd")]
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402

out = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
cfg_id = int(os.environ.get("CONFIG", "1"))
p, cfg, enc, sd, batches, docs = bench.build_model_and_batches(cfg_id, 4, 0, torch.device("cuda", 0), "")
first = {}
bad = 0
for r in range(reps):
    for i, bt in enumerate(batches):
        res = p.decode_batch_prepared(bt)
        got = dict(arc=res["arc_logits"].cpu().numpy().copy(), lab=res["label_logits"].cpu().numpy().copy(),
                   upos=res["upos_logits"].cpu().numpy().copy(), packed=res["packed"].cpu().numpy().copy())
        if r == 0:
            first[i] = got
        else:
            for k, v in got.items():
                if not np.array_equal(v, first[i][k]):
                    bad += 1
                    print("MISMATCH rep %d batch %d %s: %d elements differ" % (r, i, k, int((v != first[i][k]).sum())))
print("reps %d, mismatches %d" % (reps, bad))
np.savez(out, **{"%s_%d" % (k, i): v for i, d in first.items() for k, v in d.items()})
sys.exit(1 if bad else 0)
