"""Times the fused TT-linear chain kernel against dense F.linear at DeiT-S shapes (12 608 tokens)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dnn-compression-tensor-admm_amd"))
import torch
import torch.nn.functional as F
from tadmm import ops


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


T = 12608
out = []
for name, kin, r, nout in (("qkv", 384, 256, 1152), ("proj", 384, 256, 384), ("fc1", 384, 256, 1536), ("fc2", 1536, 256, 384)):
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(T, kin, device="cuda").to(dtype)
        win = torch.randn(r, kin, device="cuda") / kin ** 0.5
        wout = torch.randn(nout, r, device="cuda") / r ** 0.5
        wd = (wout @ win).to(dtype)
        bias = torch.randn(nout, device="cuda")
        planes = 3 if dtype == torch.float32 else 1
        wi, wo = ops.weight_planes(win, planes, 64), ops.weight_planes(wout, planes)
        row = {"layer": name, "dtype": str(dtype).split(".")[1], "dense_ms": timeit(lambda: F.linear(x, wd, bias.to(dtype)))}
        for tile in (32, 64):
            row[f"fused_tm{tile}_ms"] = timeit(ops.chain_fused(x, wi, wo, bias, nout, tile_tokens=tile, prepare_only=True))
        fl = 2.0 * T * r * (kin + nout)
        row["fused_tflops"] = fl / 1e9 / min(row["fused_tm32_ms"], row["fused_tm64_ms"])
        out.append(row)
        print(json.dumps(row), flush=True)
