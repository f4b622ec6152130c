#!/bin/bash
# usage: bash scripts/copy_evidence.sh [tag]   -- copies the judged summaries of the last r03_evidence.sh run from gpurun_out/ into profiles/
T=${1:-r03}
for c in resnet50_tt resnet18_tt deit_small_tt resnet32_tk; do cp gpurun_out/${T}_bench_$c.json profiles/${T}_bench_$c.json; done
cp gpurun_out/prof_$T/summary.md profiles/${T}_rocprofv3_summary.md
cp gpurun_out/prof_$T/pmc_traffic.json profiles/${T}_pmc_traffic.json
cp gpurun_out/prof_$T/bench_under_trace.json profiles/${T}_bench_under_trace.json
cp "$(find gpurun_out/prof_$T/trace -name '*kernel_stats.csv' | head -1)" profiles/${T}_kernel_stats.csv
for c in resnet18_tt deit_small_tt resnet32_tk; do cp gpurun_out/prof_${T}_$c/summary.md profiles/${T}_rocprofv3_summary_$c.md; done
