#!/bin/bash
# scripts/ab_kernel.sh PATTERN NAME1 NAME2 ...: serial rocprofv3 trace of the bench per variant library, average duration of the kernels matching PATTERN
pat=$1; shift
cd /tmp && export TMPDIR=/tmp
export CUBE_BENCH_SERIAL=1
for n in "$@"; do
  cp $GRAFT_REPO_ROOT/ab_tmp/lib_$n.so $GRAFT_REPO_ROOT/nlp-cube_amd/cube_amd/lib/libcubehip.so || exit 1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/abk_$n -o $n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-roofline-pass > $GRAFT_REPO_ROOT/gpurun_out/abk_$n.log 2>&1 || exit 1
  python3 - "$GRAFT_REPO_ROOT/gpurun_out/abk_$n/${n}_results.db" "$pat" "$n" <<PY
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, end - start from kernels").fetchall()
import collections
acc = collections.defaultdict(list)
for n, d in rows:
    if sys.argv[2] in n:
        acc[n.split("(")[0][-50:]].append(d)
for k, v in acc.items():
    print("%-10s %-52s %5d launches  %8.1f us avg  (sum per 20 steps... %.1f us/step)" % (sys.argv[3], k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3 / 23))
PY
done
