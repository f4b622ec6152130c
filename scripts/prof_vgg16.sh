#!/bin/bash
# usage (GPU box): bash scripts/prof_vgg16.sh  -- kernel stats of ADMM.update on the VGG-16 BN Tucker table
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_vgg16
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 $R/scripts/time_vgg16.py vgg16_bn_tk > $OUT/log 2> $OUT/err
cat $OUT/log
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/t/**/*kernel_stats.csv', recursive=True)[0]
for i, r in enumerate(csv.DictReader(open(f))):
    if i < 14: print("%-60s calls %6s total %8.2f ms avg %7.2f us" % (r['Name'][:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e3))
PY
find $OUT -name "*kernel_trace.csv" -size +2M -delete
