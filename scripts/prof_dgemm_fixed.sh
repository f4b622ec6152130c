#!/bin/bash
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/dgfixed
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT -o t -- python3 $R/scripts/dgemm_fixed_cost.py > $OUT/log 2> $OUT/err
python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'dgemm_nt_tile' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
Ks=(32,64,128,256,512,1024)
for i,K in enumerate(Ks):
    seg=d[6*i:6*i+6]
    print("K=%d: %s us" % (K, ["%.1f"%x for x in seg]))
PY
