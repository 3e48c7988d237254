#!/bin/bash
# HBM traffic of the char-conv GEMM per launch: two separate PMC passes (FETCH_SIZE, WRITE_SIZE) over a
# short bench run, as MI355X_MICROARCH.md prescribes.  scripts/pmc_traffic.sh <name>
# -> gpurun_out/pmc_<name>_{fetch,write}/ ; summarise with scripts/pmc_summary.py
name=$1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  tag=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  timeout -s KILL 240 rocprofv3 --pmc $c -d $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_$tag -o $name -f csv -- \
    python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e > $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_$tag.log 2>&1 || exit 1
done
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_fetch $GRAFT_REPO_ROOT/gpurun_out/pmc_${name}_write
