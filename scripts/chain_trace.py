"""One layer's chain under the kernel tracer: python3 scripts/chain_trace.py run <config> <layer> ; ... post <dir>

`run` projects ONE layer of a configuration alone (its own single-layer plan) a few times -- meant to be the program
after `rocprofv3 --kernel-trace --output-format csv -d DIR --`.  `post DIR` prints the kernel sequence of the last
iteration (name, duration, gap to the previous kernel) and totals per kernel name.
"""
import collections
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
sys.path.insert(0, ROOT)


def run(config, layer, iters=6):
    import torch
    import bench
    from tadmm import ops, workloads
    dev = torch.device("cuda:0")
    model, hp, fmt = workloads.build(config, seed=0)
    entries, names = bench.layer_entries(model, hp, fmt, dev)
    idx = [i for i, n in enumerate(names) if layer in n]
    ents = []
    for i in idx[:1] if not layer.endswith("*") else idx:
        e = dict(entries[i])
        e["U"] = torch.zeros_like(e["W"])
        e["Z"] = torch.empty_like(e["W"])
        ents.append(e)
    pl = ops.ProjectionPlan(ents)
    import time
    for k in range(iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pl.run(update_u=True)
        torch.cuda.synchronize()
        print("iter", k, "ms", 1e3 * (time.perf_counter() - t0), flush=True)
    print("filter", pl.filter_stats())
    pl.close()


def norm(n):
    n = n.replace("(anonymous namespace)::", "").replace("tadmm::", "").replace("void ", "")
    return re.split(r"[<(]", n)[0].strip()


def post(d, full=True):
    f = glob.glob(os.path.join(d, "**/*kernel_trace.csv"), recursive=True)[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), norm(r["Kernel_Name"]),
             int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 1) or 1)) for r in csv.DictReader(open(f))]
    rows.sort()
    starts = [i for i, k in enumerate(rows) if k[2] == "unfold_kernel"]
    it = [i for j, i in enumerate(starts) if j == 0 or rows[i - 1][2] != "unfold_kernel"]
    g = rows[it[-1]:]
    # cut at the last fold_update_kernel / resid_reduce
    end = max(i for i, k in enumerate(g) if k[2] in ("fold_update_kernel", "resid_reduce_kernel"))
    g = g[:end + 1]
    wall = (g[-1][1] - g[0][0]) / 1e3
    busy = sum(e - s for s, e, *_ in g) / 1e3
    print("last iteration: %d launches, wall %.0f us, busy (sum of durations) %.0f us" % (len(g), wall, busy))
    per = collections.defaultdict(lambda: [0, 0.0])
    for s, e, n, *_ in g:
        per[n][0] += 1
        per[n][1] += (e - s) / 1e3
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("  %-28s %4d launches %8.1f us  avg %6.1f" % (n, c, t, t / c))
    if full:
        prev_end = g[0][0]
        run_name, run_n, run_t = None, 0, 0.0
        t_rel = 0.0
        for s, e, n, gs, ws in g:
            gap = (s - prev_end) / 1e3
            prev_end = max(prev_end, e)
            if n == run_name and n.startswith("jacobi_tick"):
                run_n += 1
                run_t += (e - s) / 1e3
                continue
            if run_name:
                print("    ... x%d %s total %.1f us" % (run_n, run_name, run_t)) if run_n > 1 else None
            run_name, run_n, run_t = n, 1, (e - s) / 1e3
            print("  t=%7.1f  %-28s %6.1f us  gap %5.1f  wgs %d" % ((s - g[0][0]) / 1e3, n, (e - s) / 1e3, gap, gs // max(1, ws)))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2], sys.argv[3])
    else:
        post(sys.argv[2])
