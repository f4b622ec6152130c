"""Host and device time per call of the DeiT-small fc2 layer in bf16 inference (GPU box): F.linear on the dense weight,
TTLinearM (fast path), the cached closure alone and the cache key alone."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
import torch.nn.functional as F
from tadmm import tt_layers, functional as HF, hp as HPM
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
hp = HPM.fresh_table("tt_deit_small_patch16_224_hp.HyperParamsDictRatio2x")
lin = tt_layers.TTLinearM(1536, 384, bias=True, hp_dict=hp, name="blocks.1.mlp.fc2.weight").to(dev)
x = torch.randn(64, 197, 1536, generator=g).to(dev).to(torch.bfloat16)
wd = torch.randn(384, 1536, generator=g).to(dev).to(torch.bfloat16)
bd = torch.randn(384, generator=g).to(dev).to(torch.bfloat16)

def dev_time(fn, iters=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

def host_time(fn, iters=2000):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters): fn()
    dt = (time.perf_counter() - t) / iters * 1e6
    torch.cuda.synchronize()
    return dt

with torch.no_grad():
    lin(x)
    c = lin.__dict__["_chain_cache"]
    fast = c["fast"]
    print("device us/call: F.linear dense %.1f | module %.1f | closure %.1f" % (dev_time(lambda: F.linear(x, wd, bd)), dev_time(lambda: lin(x)), dev_time(lambda: fast(x))))
    print("host   us/call (queue never waited for): key %.2f" % host_time(lambda: HF.param_key(*lin.tt_cores, lin.bias), 20000))
    small = x[:1, :4]
    print("host   us/call on a tiny input (GPU never the bound): F.linear %.1f | module %.1f | closure %.1f | nn.Identity module call %.2f" % (
        host_time(lambda: F.linear(small, wd, bd)), host_time(lambda: lin(small)), host_time(lambda: fast(small)), host_time(lambda: torch.nn.Identity()(small))))
    x2 = x.reshape(-1, 1536)
    print("device us/call 2-D input: F.linear %.1f | closure %.1f ; dense weight contiguous %s stride %s dtype %s" % (dev_time(lambda: F.linear(x2, wd, bd)), dev_time(lambda: fast(x2)), c["dense"].is_contiguous(), c["dense"].stride(), c["dense"].dtype))

with torch.no_grad():
    for i in range(4):
        c = lin.__dict__.get("_chain_cache")
        k = HF.param_key(*lin.tt_cores, lin.bias)
        print("call", i, "cache" if c is not None else "no cache", "fast_key match", None if c is None else c.get("fast_key") == k,
              "dtype", None if c is None else c.get("fast_dtype"), "dev", None if c is None else (c.get("fast_dev"), x.device, c.get("fast_dev") == x.device))
        torch.cuda.synchronize(); t = time.perf_counter(); lin(x); torch.cuda.synchronize(); print("   wall us", (time.perf_counter() - t) * 1e6)
    print("device us/call again: module %.1f closure %.1f" % (dev_time(lambda: lin(x)), dev_time(lambda: lin.__dict__["_chain_cache"]["fast"](x))))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(200): lin(x)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
