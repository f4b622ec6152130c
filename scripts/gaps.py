"""Inter-kernel gaps from a rocprofv3 --kernel-trace csv: python scripts/gaps.py <dir>"""
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**/*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("tadmm::", "")) for r in csv.DictReader(open(f))]
rows.sort()
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap[(n0[:24], n1[:24])].append(s1 - e0); dur[n0[:24]].append(e0 - s0)
print("total span ms", (rows[-1][1] - rows[0][0]) / 1e6, "sum kernel ms", sum(e - s for s, e, _ in rows) / 1e6)
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:10]:
    v2 = sorted(v); print("gap %-26s -> %-26s n=%5d  median %.2f us  mean %.2f us  p90 %.2f  total %.2f ms" % (k[0], k[1], len(v), v2[len(v)//2] / 1e3, sum(v) / len(v) / 1e3, v2[int(len(v)*0.9)] / 1e3, sum(v) / 1e6))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:6]:
    v2 = sorted(v); print("dur %-26s n=%5d median %.2f us mean %.2f us total %.2f ms" % (k, len(v), v2[len(v)//2] / 1e3, sum(v) / len(v) / 1e3, sum(v) / 1e6))
if len(sys.argv) > 2:
    pat = sys.argv[2]
    sel = [(s, e, n) for s, e, n in rows if pat in n]
    print("per-launch durations (us) of", pat, ":", " ".join("%.1f" % ((e - s) / 1e3) for s, e, n in sel[-16:]))
