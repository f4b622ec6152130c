#!/bin/bash
# usage (GPU box): bash scripts/r03_ab.sh  -- A/B of the guard + first run of the direct small solver
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py tests/test_gpu_projection.py tests/test_gpu_round2.py -m gpu -x -q > gpurun_out/r03_ab_tests.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/r03_ab_tests.log
for g in 4 0; do
  TADMM_FILTER_GUARD=$g python bench.py --no-cpu-baseline --no-forward --no-per-layer --no-roofline > gpurun_out/r03_ab_g$g.json 2> gpurun_out/r03_ab_g$g.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r03_ab_g$g.json")); print("guard=$g ms_per_step", d["ms_per_step"])
PY
done
for sd in 1 0; do
  TADMM_SMALL_DIRECT=$sd python bench.py --config resnet32_tk --no-cpu-baseline --no-per-layer > gpurun_out/r03_ab_tk$sd.json 2> gpurun_out/r03_ab_tk$sd.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r03_ab_tk$sd.json")); print("direct=$sd tk ms_per_step", d["ms_per_step"], d["hooi_sweeps"]["max"], d.get("roofline", {}).get("avg_launch_us"), d.get("roofline", {}).get("launches_per_step"))
PY
done
bash scripts/chain_trace.sh resnet50_tt layer4.1.conv2 chain_l4b > /dev/null; head -32 gpurun_out/chain_l4b/seq.txt
