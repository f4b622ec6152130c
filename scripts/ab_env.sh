#!/bin/bash
# usage (GPU box): bash scripts/ab_env.sh VAR=val [VAR=val ...] -- one bench line per setting (each argument is one run)
for kv in "$@"; do
  echo -n "$kv : "
  env $kv timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-forward --no-cpu-baseline --no-per-layer --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('phases_ms',{}).get('jacobi_sweeps'), d.get('residual_sq'))"
done
