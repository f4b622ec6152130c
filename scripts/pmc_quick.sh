#!/bin/bash
# usage: bash scripts/pmc_quick.sh <tag> "<counters>"   -- per-kernel PMC averages of a short bench run (GPU box)
TAG=${1:-p}; CNT=${2:-FETCH_SIZE}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/run -- python3 $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-roofline > /dev/null 2> $OUT/err.log
cd $R && python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, os
out = sys.argv[1]
f = glob.glob(os.path.join(out, "run/**/*counter_collection.csv"), recursive=True)
if not f: print("no counter csv"); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for row in csv.DictReader(open(f[0])):
    name = row["Kernel_Name"].split("(")[0].replace("tadmm::", "").replace("void ", "")[:40]
    a = acc[name][row["Counter_Name"]]; a[0] += 1; a[1] += float(row["Counter_Value"])
for name, cs in sorted(acc.items(), key=lambda kv: -sum(v[1] for v in kv[1].values()))[:10]:
    print(name, {c: "%.4g avg (%d)" % (v[1] / v[0], v[0]) for c, v in cs.items()})
PY
find $OUT -name "*.csv" -size +3M -delete
