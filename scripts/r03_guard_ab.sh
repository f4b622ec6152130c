#!/bin/bash
# separate processes, alternating, three rounds: guard on a side stream / in line / off
for rnd in 1 2 3; do
  for s in "TADMM_X=1" "TADMM_FILTER_GUARD=3" "TADMM_FILTER_GUARD=0"; do
    echo -n "$s : "
    env $s python bench.py --no-cpu-baseline --no-forward --no-per-layer --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"
  done
done
