"""Strided fp32 GEMM of the library (tadmm_gemm: v_mfma_f32_32x32x2_f32 tile) at plan-like and large shapes, vs torch.mm.
A three-plane bf16 version of this tile (operands split on the way into LDS) was measured in round 2: 66 vs 91 TF/s at
4096^3 and 2x slower inside the plan (the split is redone by every tile that reuses an operand), so it was not kept."""
import os, sys, subprocess, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
if len(sys.argv) > 1:
    sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
    import torch
    from tadmm import ops
    torch.manual_seed(0)
    out = {}
    for M, N, K, ta, tb in ((4096, 4096, 4096, False, False), (256, 2048, 1152, True, False), (1152, 256, 2048, False, True),
                            (480, 4608, 105, False, False), (12608, 1152, 384, False, True), (64, 64, 64, False, False)):
        a = torch.randn(K, M, device="cuda").t() if ta else torch.randn(M, K, device="cuda")
        b = torch.randn(N, K, device="cuda").t() if tb else torch.randn(K, N, device="cuda")
        ref = a.double() @ b.double()
        c = ops.mm(a, b)
        err = ((c.double() - ref).abs().max() / ref.abs().max()).item()
        for _ in range(3): ops.mm(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): ops.mm(a, b)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        e0.record()
        for _ in range(n): torch.mm(a, b)
        e1.record(); torch.cuda.synchronize()
        tms = e0.elapsed_time(e1) / n
        out["%dx%dx%d%s%s" % (M, N, K, "T" if ta else "N", "T" if tb else "N")] = dict(
            ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1), err=float("%.2e" % err), torch_ms=round(tms, 4))
    print(sys.argv[1], json.dumps(out), flush=True)
else:
    subprocess.run([sys.executable, __file__, "tadmm_gemm"], check=True)
