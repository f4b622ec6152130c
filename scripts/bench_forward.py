"""Forward throughput of the factorised layers against the dense layers they replace:
python scripts/bench_forward.py  -> one JSON object per layer/dtype (tadmm/fwdbench.py; the same rows go into the
`forward` block of bench.py's line)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
from tadmm import fwdbench  # noqa: E402

if __name__ == "__main__":
    for row in fwdbench.run():
        print(json.dumps(row), flush=True)
