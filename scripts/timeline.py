"""Steady-state timeline of a rocprofv3 --kernel-trace of bench.py (rocpd .db): step time, busy fraction, kernel
time per stream and the kernels of one step with their start offsets (which stage overlaps which).
    python scripts/timeline.py gpurun_out/prof_x/x_results.db"""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n); return n.split("(")[0][:60]
# find steady-state: last 10 occurrences of mst_select_kernel delimit steps
idx = [i for i,r in enumerate(rows) if 'mst_select' in r[0]]
print(len(rows), len(idx))
t0 = rows[idx[-6]][2]; t1 = rows[idx[-1]][2]
sel = [r for r in rows if r[1] >= t0 and r[2] <= t1]
print("5 steps span ms", (t1-t0)/1e6, "per step", (t1-t0)/5e6)
# union busy
busy=0; cs=ce=None
for n,s,e,*_ in sel:
    if ce is None or s>ce:
        if ce is not None: busy+=ce-cs
        cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("busy frac", busy/(t1-t0))
streams = sorted(set(r[3] for r in sel))
for st in streams:
    ss=[r for r in sel if r[3]==st]
    print("stream", st, "kernels", len(ss), "sum ms/step", sum(r[2]-r[1] for r in ss)/5e6)
# print one step timeline
a = rows[idx[-3]][2]; b = rows[idx[-2]][2]
for n,s,e,st,q in rows:
    if s>=a and s<b:
        print("%8.1f %8.1f  st%s %s" % ((s-a)/1e3, (e-s)/1e3, st, short(n)))
