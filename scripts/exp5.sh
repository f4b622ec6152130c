python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for mode in 2 1; do
echo "mode $mode"
TADMM_JACOBI_MODE=$mode TADMM_DEBUG=1 timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "tadmm\] step|ms_per_step" | cut -c1-300 | tail -5
done
