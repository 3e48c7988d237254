#!/bin/bash
# rocprofv3 kernel trace of the end-to-end process(Document) run: scripts/prof_e2e.sh <name> [batch size]
name=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$name -o $name -- python3 $GRAFT_REPO_ROOT/scripts/e2e_profile.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log 2>&1
grep process $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log
