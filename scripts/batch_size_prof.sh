#!/bin/bash
# scripts/batch_size_prof.sh MODE B...   -> gpurun_out/bsp_<MODE>_<B>/ (rocprofv3 kernel trace of scripts/batch_size_prof.py B 24 MODE)
mode=$1; shift
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/bsp_${mode}_$B -o bsp -- python3 $GRAFT_REPO_ROOT/scripts/batch_size_prof.py $B 24 $mode > $GRAFT_REPO_ROOT/gpurun_out/bsp_${mode}_$B.log 2>&1 || exit 1
  tail -1 $GRAFT_REPO_ROOT/gpurun_out/bsp_${mode}_$B.log
done
