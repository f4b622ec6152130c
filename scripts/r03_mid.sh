#!/bin/bash
# GPU box: parity of the direct Rayleigh-Ritz route, then its A/B against the tournament
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_filter.py tests/test_gpu_projection.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/mid_tests.log 2>&1; echo "tests exit $?"; tail -5 gpurun_out/mid_tests.log
bash scripts/ab_procs.sh resnet50_tt "TADMM_MID_DIRECT=1" "TADMM_MID_DIRECT=0" 2>&1 | tee gpurun_out/mid_ab.log
bash scripts/ab_procs.sh resnet18_tt "TADMM_MID_DIRECT=1" "TADMM_MID_DIRECT=0" 2>&1 | tee -a gpurun_out/mid_ab.log
