#!/bin/bash
# usage (GPU box): bash scripts/r03_check.sh <tag>  -- whole GPU suite, then a short bench line
TAG=${1:-t}
python -m pytest tests -m gpu -x -q > gpurun_out/r03_${TAG}_tests.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/r03_${TAG}_tests.log
python bench.py --no-cpu-baseline --no-forward --no-per-layer > gpurun_out/r03_${TAG}_bench.json 2> gpurun_out/r03_${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/r03_${TAG}_bench.json"))
print("ms_per_step", d["ms_per_step"], d.get("filter"))
PY
