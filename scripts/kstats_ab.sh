#!/bin/bash
# usage (GPU box): bash scripts/kstats_ab.sh <config> "<VAR=val ...>" ...   -- launches and total time per kernel for each setting
CFG=$1; shift
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for s in "$@"; do
  i=$((i+1))
  OUT=$GRAFT_REPO_ROOT/gpurun_out/kstats_$i
  rm -rf $OUT
  env $s rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-per-layer --no-forward --no-roofline > $OUT.json 2> $OUT.err
  python3 - "$s" $OUT <<'PY'
import csv, glob, sys, json
s, out = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ms = json.loads(open(out + ".json").read().strip().splitlines()[-1])["ms_per_step"]
print("==", s, "ms/step under trace %.2f" % ms)
for r in rows[:9]:
    print("   %-44s calls/step %7.1f  ms/step %6.3f  avg us %6.1f" % (r["Name"].replace("tadmm::", "").replace("void ", "")[:44], int(r["Calls"]) / 8.0, float(r["TotalDurationNs"]) / 8e6, float(r["AverageNs"]) / 1e3))
PY
  find $OUT -name "*kernel_trace.csv" -delete
done
