import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))

import torch
from tadmm import ops
T, kin, r, nout = 12608, 384, 256, 1152
for dtype in (torch.float32, torch.bfloat16):
    x = torch.randn(T, kin, device="cuda").to(dtype)
    win = torch.randn(r, kin, device="cuda") / kin ** 0.5
    wout = torch.randn(nout, r, device="cuda") / r ** 0.5
    planes = 3 if dtype == torch.float32 else 1
    wi, wo = ops.weight_planes(win, planes, 64), ops.weight_planes(wout, planes)
    for tile in (32, 64):
        go = ops.chain_fused(x, wi, wo, None, nout, tile_tokens=tile, prepare_only=True)
        for _ in range(3):
            go()
        torch.cuda.synchronize()
        print(dtype, tile, file=sys.stderr, flush=True)
        os.environ["TADMM_CHAIN_STAMPS_DUMP"] = "1"
        go()          # dumps the stamps of the previous launch
        del os.environ["TADMM_CHAIN_STAMPS_DUMP"]
