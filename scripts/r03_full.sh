#!/bin/bash
# GPU box: the whole GPU suite in one process, log under gpurun_out/
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/full_gpu_tests.log 2>&1; rc=$?
tail -6 gpurun_out/full_gpu_tests.log; echo "pytest rc=$rc"; exit $rc
