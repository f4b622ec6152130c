#!/bin/bash
# ms/step of every TT benchmark configuration (no CPU baseline)
for c in resnet50_tt resnet18_tt deit_small_tt; do
  python bench.py --config $c --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline > /tmp/c.json 2> /tmp/c.err || { tail -5 /tmp/c.err; exit 1; }
  python - $c <<'PY'
import json, sys
d = json.loads(open('/tmp/c.json').read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step %.2f" % d["ms_per_step"], "it/s %.1f" % d["value"], "svd GFLOP/s %.0f" % d["svd_gflops_per_s"], "phases", {k: round(v, 2) for k, v in d["phases_ms"].items()}, "roofline frac", (d.get("roofline") or {}).get("frac"))
PY
done
