set -e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/t2.log; cat gpurun_out/t2.log
for inner in 1 2 3; do for tol in 1e-9 1e-6; do
  echo "inner=$inner tol=$tol"
  TADMM_JACOBI_INNER=$inner TADMM_JACOBI_TOL=$tol timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2),'ms', {k:round(v,2) for k,v in d['phases_ms'].items()})"
done; done
