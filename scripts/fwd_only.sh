#!/bin/bash
# GPU box: the forward block of bench.py alone (tadmm/fwdbench.py), one line per row
python - <<'PY'
import sys, os, json
sys.path.insert(0, "dnn-compression-tensor-admm_amd")
import torch
from tadmm import fwdbench
for r in fwdbench.run(torch.device("cuda:0")):
    print(r["layer"][:52], r["dtype"], "eager", r["ms"], r["dense_ms"], r["eager_speedup_vs_dense"], "| graph", r.get("graph_ms"), r.get("dense_graph_ms"), "->", r["speedup_vs_dense"], "|", r.get("path", "")[:30])
PY
