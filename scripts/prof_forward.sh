#!/bin/bash
# usage: bash scripts/prof_forward.sh   -- kernel-trace stats of scripts/bench_forward.py (GPU box)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fwdprof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o fwd -- python3 $R/scripts/bench_forward.py > $OUT/bench.log 2> $OUT/trace.err
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f))):
    n=r['Name']
    if 'tadmm' in n: print(n[:150], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
