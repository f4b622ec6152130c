#!/bin/bash
# usage (GPU box): bash scripts/r03_tri.sh -- stamps + tests + Tucker timing of the direct small solver
TADMM_LIB=libtadmm_hip_stamp.so python scripts/stamp_tri.py 2>&1 | grep -v amdgpu.ids | tail -16
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_layers.py tests/test_gpu_projection.py tests/test_gpu_round2.py tests/test_gpu_filter.py -m gpu -x -q > gpurun_out/r03_tri_tests.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/r03_tri_tests.log
for sd in 1 0; do
  TADMM_SMALL_DIRECT=$sd python bench.py --config resnet32_tk --no-cpu-baseline --no-per-layer > gpurun_out/r03_tri_tk$sd.json 2> gpurun_out/r03_tri_tk$sd.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r03_tri_tk$sd.json")); print("direct=$sd tk ms_per_step", d["ms_per_step"], d["hooi_sweeps"]["max"], d.get("phases_ms"))
PY
done
for g in 3 0; do
  TADMM_FILTER_GUARD=$g python bench.py --no-cpu-baseline --no-forward --no-per-layer --no-roofline > gpurun_out/r03_tri_g$g.json 2> gpurun_out/r03_tri_g$g.err
  python - <<PY
import json
d = json.load(open("gpurun_out/r03_tri_g$g.json")); print("guard=$g ms_per_step", d["ms_per_step"])
PY
done
