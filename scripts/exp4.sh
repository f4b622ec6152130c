mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for ns in 0 1; do
if [ $ns = 1 ]; then export TADMM_NO_SUPER=1; fi
TADMM_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "tadmm|ms_per_step" | cut -c1-330 | tail -5
done
