#!/bin/bash
mkdir -p gpurun_out
TADMM_MID_DEBUG=1 python bench.py --config resnet50_tt --steps 2 --warmup 1 --no-cpu-baseline --no-forward --no-per-layer --no-roofline > gpurun_out/mid_dbg.log 2>&1
sort gpurun_out/mid_dbg.log | uniq -c | sort -rn | head -20
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/prof_mid -o mid -- python3 /root/repo/bench.py --config resnet50_tt --steps 5 --warmup 2 --no-cpu-baseline --no-forward --no-per-layer --no-roofline > /root/repo/gpurun_out/mid_prof.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/prof_mid/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:14]:
    print(r['Name'][:70], r['Calls'], r['TotalDurationNs'], r['AverageNs'])
PY
