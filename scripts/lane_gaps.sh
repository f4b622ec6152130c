#!/bin/bash
# usage (GPU box): bash scripts/lane_gaps.sh   -- per device queue: busy time, idle gaps and launch count of one iteration
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lane_gaps
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/t -- python3 $R/bench.py --config ${CFG:-resnet50_tt} --steps 6 --warmup 3 --no-cpu-baseline --no-roofline --no-per-layer --no-forward > $OUT/bench.json 2> $OUT/err
python3 - <<PY
import csv, glob, collections, re
f = glob.glob('$OUT/t/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
def norm(n):
    n = n.replace("(anonymous namespace)::", "").replace("tadmm::", "").replace("void ", "")
    return re.split(r"[<(]", n)[0].strip()
byq = collections.defaultdict(list)
for r in rows:
    byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), norm(r['Kernel_Name'])))
for q, ks in sorted(byq.items()):
    ks.sort()
    if len(ks) < 500: continue
    # the last 6 iterations: split the queue's timeline at unfold_kernel launches (first kernel of an iteration)
    starts = [i for i, k in enumerate(ks) if k[2] == 'unfold_kernel']
    # iterations begin where an unfold follows a fold_update
    it = [i for j, i in enumerate(starts) if j == 0 or ks[i - 1][2] != 'unfold_kernel']
    groups = []
    for a, b in zip(it[:-1], it[1:]):
        groups.append(ks[a:b])
    groups = [g for g in groups if len(g) > 100][-5:]
    for g in groups[-2:]:
        wall = (g[-1][1] - g[0][0]) / 1e3
        busy = sum(e - s for s, e, _ in g) / 1e3
        gaps = [(g[i + 1][0] - g[i][1]) / 1e3 for i in range(len(g) - 1)]
        big = sorted(((gp, g[i][2], g[i + 1][2]) for i, gp in enumerate(gaps)), reverse=True)[:8]
        print("queue %s: %d launches, wall %.0f us, busy %.0f us, gaps %.0f us (median %.2f us)" % (q, len(g), wall, busy, sum(gaps), sorted(gaps)[len(gaps) // 2]))
        print("   largest gaps:", ["%.0f us after %s before %s" % b for b in big])
        per = collections.defaultdict(lambda: [0, 0.0])
        for s, e, n in g:
            per[n][0] += 1; per[n][1] += (e - s) / 1e3
        print("   ", sorted(((round(v[1]), v[0], k) for k, v in per.items()), reverse=True)[:8])
PY
