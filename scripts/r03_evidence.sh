#!/bin/bash
# usage (GPU box): bash scripts/r03_evidence.sh <commit>  -- the round's evidence in one call: tests, profiles, bench lines
C=${1:-unknown}
python -m pytest tests -m gpu -q > gpurun_out/r03_final_tests.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r03_final_tests.log
bash scripts/profile.sh r03 $C resnet50_tt > gpurun_out/r03_profile.log 2>&1; tail -5 gpurun_out/r03_profile.log
bash scripts/all_configs.sh r03
for c in resnet18_tt deit_small_tt resnet32_tk; do
  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r03_$c/trace -- python3 $GRAFT_REPO_ROOT/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline --no-per-layer --no-forward > $GRAFT_REPO_ROOT/gpurun_out/prof_r03_$c/bench_under_trace.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r03_$c/trace.err
  cd $GRAFT_REPO_ROOT && python3 scripts/summarize_profile.py gpurun_out/prof_r03_$c $C $c > gpurun_out/prof_r03_$c/summary.md 2>/dev/null
  find gpurun_out/prof_r03_$c -name "*kernel_trace.csv" -size +2M -delete
  head -16 gpurun_out/prof_r03_$c/summary.md | tail -10
done
