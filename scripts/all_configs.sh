#!/bin/bash
# One full bench line (roofline, roofline_sweep, cpu_baseline, per_layer_svd_gflops) per BASELINE configuration:
#   bash scripts/all_configs.sh [tag]      (GPU box)   ->  gpurun_out/<tag>_bench_<config>.json, one summary line each
# The files are copied to profiles/ by hand after the run (gpurun_out/ is scratch).
TAG=${1:-r03}
for c in resnet50_tt resnet18_tt deit_small_tt resnet32_tk; do
  extra=""
  [ "$c" != "resnet50_tt" ] && extra="--no-forward"
  python bench.py --config $c $extra > gpurun_out/${TAG}_bench_$c.json 2> gpurun_out/${TAG}_bench_$c.err || { tail -5 gpurun_out/${TAG}_bench_$c.err; exit 1; }
  python - $c gpurun_out/${TAG}_bench_$c.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(sys.argv[1], "ms/step %.2f" % d["ms_per_step"], "it/s %.1f" % d["value"], "| roofline", (r.get("kernel") or "")[:28], "frac %.4f" % r.get("frac", 0.0),
      "| sweep frac %.4f" % d["roofline_sweep"]["frac"], "| cpu %.3f it/s on %s thread(s)" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"]),
      "| filter", d.get("filter"))
PY
done
