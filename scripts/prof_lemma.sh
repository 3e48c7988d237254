#!/bin/bash
# rocprofv3 kernel trace of scripts/lemma_profile.py: scripts/prof_lemma.sh <name>
name=$1; shift
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$name -o $name -- python3 $GRAFT_REPO_ROOT/scripts/lemma_profile.py > $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log 2>&1
head -n 3 $GRAFT_REPO_ROOT/gpurun_out/prof_$name.log
