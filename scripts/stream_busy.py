"""Per-stream busy time of a rocprofv3 kernel trace (rocpd .db) over the steady-state part of a bench run:
    python scripts/stream_busy.py x_results.db [skip_fraction]"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id from kernels order by start").fetchall()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
t_lo = rows[0][1] + skip * (rows[-1][2] - rows[0][1])
sel = [r for r in rows if r[1] >= t_lo]
span = sel[-1][2] - sel[0][1]
print("window %.2f ms, %d kernels" % (span / 1e6, len(sel)))
per = collections.defaultdict(list)
for n, s, e, st in sel:
    per[st].append((s, e, n))
for st, v in sorted(per.items()):
    busy = sum(e - s for s, e, _ in v)
    acc = collections.defaultdict(lambda: [0, 0])
    for s, e, n in v:
        k = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "").split("(")[0][:60]
        acc[k][0] += e - s
        acc[k][1] += 1
    print("stream %s: busy %.2f ms = %.0f %% of the window" % (st, busy / 1e6, 100.0 * busy / span))
    for k, (t, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:8]:
        print("    %-62s %5d  %8.2f ms  %7.1f us each" % (k, c, t / 1e6, t / c / 1e3))
