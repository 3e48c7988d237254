"""python scripts/npz_equal.py a.npz b.npz -> exit 0 when every array is bitwise equal."""
import sys
import numpy as np
a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    if not np.array_equal(a[k], b[k]):
        bad += 1
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print("DIFF", k, int((a[k] != b[k]).sum()), "max abs", d.max())
print("arrays %d, differing %d" % (len(a.files), bad))
sys.exit(1 if bad else 0)
