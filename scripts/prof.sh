#!/bin/bash
# rocprofv3 kernel trace of the default bench on the GPU box: scripts/prof.sh <name> [bench args]
# (run through gpurun; the summary is made afterwards by scripts/prof_summary.py)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-roofline-pass "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log 2>&1
grep metric $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log | cut -c1-160
