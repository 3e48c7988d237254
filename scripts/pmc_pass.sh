#!/bin/bash
# One rocprofv3 --pmc pass over a short SERIAL bench run (kernels never overlap): scripts/pmc_pass.sh <name> <counter> [counter ...]
# (at most 8 SQ_, 4 TCC_, 2 GRBM_ counters per pass; FETCH_SIZE and WRITE_SIZE each in a pass of their own)
# -> gpurun_out/pmc_<name>/ ; summarise with scripts/pmc_mean.py
name=$1; shift
cd /tmp && export TMPDIR=/tmp
export CUBE_BENCH_SERIAL=1
timeout -s KILL 240 rocprofv3 --pmc "$@" -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$name -o $name -f csv -- \
  python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/pmc_$name.log 2>&1 || exit 1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$name
