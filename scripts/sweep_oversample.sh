#!/bin/bash
# usage (GPU box): bash scripts/sweep_oversample.sh   -- iteration time against the filter's oversampling factor
for o in 1.3 1.42 1.55 1.7 1.85 2.05; do
  echo -n "oversample $o: "
  TADMM_FILTER_OVERSAMPLE=$o timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-forward --no-cpu-baseline --no-per-layer --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('phases_ms',{}).get('jacobi_sweeps'))"
done
