#!/bin/bash
# usage: bash scripts/prof_quick.sh <tag> [bench args]   -- kernel-trace stats of a short bench run (GPU box)
TAG=${1:-q}; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench_under_trace.json 2> $OUT/trace.err
cd $R && python3 scripts/summarize_profile.py $OUT > $OUT/summary.md 2> $OUT/summary.err; head -40 $OUT/summary.md
find $OUT -name "*kernel_trace.csv" -size +2M -delete
