#!/bin/bash
# rocprofv3 kernel trace of the bench on ONE stream (kernels never overlap): scripts/prof_serial.sh <name> [bench args]
# environment variables of the caller (e.g. A/B switches) are passed through
name=$1; shift
cd /tmp && export TMPDIR=/tmp
export CUBE_BENCH_SERIAL=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log 2>&1
grep metric $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log | cut -c1-160
