#!/bin/bash
# usage: bash scripts/prof_chain.sh   -- kernel-trace stats of scripts/bench_chain.py (GPU box)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/chainprof
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o chain -- python3 $R/scripts/bench_chain.py > $OUT/bench.log 2> $OUT/trace.err
find $OUT -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(r['Name'][:120], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
