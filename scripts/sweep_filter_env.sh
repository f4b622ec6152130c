#!/bin/bash
# usage (GPU box): bash scripts/sweep_filter_env.sh  -- iteration time against the filter's degree cap / conditioning budget
run() { echo -n "$* : "; env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-forward --no-cpu-baseline --no-per-layer --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('filter'))"; }
run TADMM_FILTER_DEGREE=8
run TADMM_FILTER_DEGREE=10
run TADMM_FILTER_DEGREE=12
run TADMM_FILTER_DEGREE=16
run TADMM_FILTER_DEGREE=12 TADMM_FILTER_COND=3e6
run TADMM_FILTER_DEGREE=16 TADMM_FILTER_COND=3e6
run TADMM_FILTER_DEGREE=8 TADMM_FILTER_COND=3e6
