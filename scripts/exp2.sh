mkdir -p gpurun_out
python -m pytest tests/test_gpu_projection.py -x -q -k properties 2>&1 | grep -E "assert|Error|error|^E" | head -20
TADMM_DEBUG=1 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "tadmm|ms_per_step" | cut -c1-400 | tail -30
