import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
from tadmm import ops
from tadmm._cabi import TadmmError
def trial(name, G):
    try:
        lam, V, sweeps = ops.eigh(G.cuda())
        ref = torch.linalg.eigvalsh(G).flip(0)
        print(G.shape[0], name, "sweeps", sweeps, "lam err %.2e" % ((lam.cpu() - ref).abs().max().item() / max(1e-300, ref.abs().max().item())), flush=True)
    except TadmmError as e:
        print(G.shape[0], name, "ERROR", str(e)[:100], flush=True)
for N in (16, 32, 128, 480, 1152):
    one = torch.ones(N, N, dtype=torch.float64)
    trial("ones", one)
    blk = torch.zeros(N, N, dtype=torch.float64); blk[:8, :8] = 1.0
    trial("8x8 block", blk)
    blk = torch.zeros(N, N, dtype=torch.float64); blk[4:12, 4:12] = 1.0; blk[4, 4] += 1e-13
    trial("8x8 straddling + jitter", blk)
    v = torch.randn(N, 3, dtype=torch.float64)
    trial("rank 3", v @ v.t())
