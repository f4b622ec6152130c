"""Phase isolation of the fused chain kernel: TADMM_CHAIN_DBG bit 1 = no Y stores, 2 = no product 2, 4 = all weight
loads hit one tile (L1-resident)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
    import torch
    from tadmm import ops
    T, kin, r, nout = 12608, 384, 256, 1152
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        x = torch.randn(T, kin, device="cuda").to(dtype)
        win = torch.randn(r, kin, device="cuda") / kin ** 0.5
        wout = torch.randn(nout, r, device="cuda") / r ** 0.5
        planes = 3 if dtype == torch.float32 else 1
        wi, wo = ops.weight_planes(win, planes, 64), ops.weight_planes(wout, planes)
        for tile in (32, 64):
            go = ops.chain_fused(x, wi, wo, None, nout, tile_tokens=tile, prepare_only=True)
            for _ in range(5):
                go()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                go()
            e1.record(); torch.cuda.synchronize()
            res[f"{str(dtype)[6:]}_tm{tile}"] = round(e0.elapsed_time(e1) / 30 * 1000, 1)
    print(sys.argv[1], json.dumps(res), flush=True)
else:
    for dbg in (0, 1, 2):
        env = dict(os.environ, TADMM_CHAIN_DBG=str(dbg))
        subprocess.run([sys.executable, __file__, f"dbg={dbg}"], env=env, check=True)
