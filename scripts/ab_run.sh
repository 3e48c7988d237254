#!/bin/bash
# A/B on ONE box: scripts/ab_run.sh NAME1 NAME2 ... -- for each ab_tmp/lib_NAME.so the serial bench's one-stream char-conv timing, twice
for rep in 1 2; do
for n in "$@"; do
  cp ab_tmp/lib_$n.so nlp-cube_amd/cube_amd/lib/libcubehip.so || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-e2e --steps 60 --warmup 10 > gpurun_out/ab_$n.json 2>/dev/null || exit 1
  python - "$n" <<PY
import json, sys
d = json.loads(open("gpurun_out/ab_%s.json" % sys.argv[1]).read().strip().splitlines()[-1])
print("%-12s value %7d  step %.3f ms  char-conv alone %.1f us (frac %.4f)  pipelined %.1f us" % (sys.argv[1], d["value"], d["ms_per_step"], 1e3 * d["roofline"]["alone"]["ms_per_launch"], d["roofline"]["alone"]["frac"], 1e3 * d["roofline"]["ms_per_launch"]))
PY
done
done
