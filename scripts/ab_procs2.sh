#!/bin/bash
# usage (GPU box): bash scripts/ab_procs2.sh <config> "<VAR=val ...>" ...  -- like ab_procs.sh, also prints the filter statistics
# (the instrumented pass of the roofline block counts eligible problems / fallbacks / stages)
CFG=$1; shift
for rnd in 1 2 3; do
  for s in "$@"; do
    echo -n "$s : "
    env $s python bench.py --config $CFG --no-cpu-baseline --no-forward --no-per-layer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d.get('filter'))"
  done
done
