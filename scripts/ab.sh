#!/bin/bash
# usage: bash scripts/ab.sh ENVVAR v1 v2 ...   -> ms/step of bench.py (headline config) for each value of ENVVAR
VAR=$1; shift
for v in "$@"; do
  env $VAR=$v python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu-baseline --no-roofline ${BENCH_ARGS} > /tmp/ab.json 2> /tmp/ab.err || { tail -5 /tmp/ab.err; exit 1; }
  python - "$VAR" "$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/ab.json').read().strip().splitlines()[-1])
print(sys.argv[1], sys.argv[2], "ms/step %.2f" % d["ms_per_step"], "sweeps", d.get("phases_ms", {}).get("jacobi_sweeps"))
PY
done
