"""Forward throughput of the factorised layers against the dense layers they replace (SURVEY.md 8d config 5,
hot path B: TTLinear.py:75-93, TTConv.py:130-153, TKConv.py:93-98).

Every row times the MODULE call (`layer(x)`, weights packed once, inference mode) and the dense torch layer of the
same shape and dtype with HIP events on torch's current stream -- the stream the chain kernels are launched on --
plus, for the chain launches, the bare kernel time (descriptor prebuilt).  `roofline` prices the kernel against the
bf16 matrix-core peak with the EXECUTED flops (6 bf16 products per fp32 product in three-plane mode) and gives the
algorithmic rate beside it.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import functional as HF
from . import hp as HPM
from . import ops, tk_layers, tt_layers

PEAK_BF16_TFLOPS = 2500.0        # MI355X dense bf16 matrix peak (MI355X_MICROARCH.md)
PEAK_F32_TFLOPS = 157.0          # fp32 matrix peak, for the algorithmic rate of the three-plane mode


def _time(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _graph_time(fn, iters):
    """ms per call with the host out of the loop: `iters` calls captured into ONE hipGraph (torch.cuda.CUDAGraph; the chain
    kernels are plain launches on torch's current stream, so they capture like any torch op) and the graph replayed.  At
    these sizes an eager call is host-bound -- ~20 us of dispatch for F.linear, ~26 us for a module call, around kernels of
    10 - 40 us -- so the eager figures compare Python overheads, not layers.  None when the capture fails."""
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(iters):
                fn()
        graph.replay()
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / (iters * reps)
    except Exception:                                   # noqa: BLE001 -- measurement helper: report "no graph figure"
        torch.cuda.synchronize()
        return None


def _pair(timer, fn_a, fn_b, arg, rounds: int = 3):
    """Best of `rounds` ALTERNATING measurements of two callables: the clocks of a freshly woken part keep rising for the
    first tens of milliseconds, so whichever side is measured first looks 3 - 5 % slower (scripts/graph_order.py)."""
    best_a = best_b = None
    for _ in range(rounds):
        a, b = timer(fn_a, arg), timer(fn_b, arg)
        if a is None or b is None:
            return None, None
        best_a = a if best_a is None else min(best_a, a)
        best_b = b if best_b is None else min(best_b, b)
    return best_a, best_b


def _row(name, dtype, ms, dense_ms, kernel_ms, alg_flops, launches, graph_ms=None, dense_graph_ms=None):
    planes = 6 if dtype == torch.float32 else 1
    row = {"layer": name, "dtype": "f32(3xbf16)" if dtype == torch.float32 else "bf16",
           "ms": round(ms, 4), "dense_ms": round(dense_ms, 4), "eager_speedup_vs_dense": round(dense_ms / ms, 3),
           "launches": launches}
    if graph_ms is not None and dense_graph_ms is not None:
        row["graph_ms"], row["dense_graph_ms"] = round(graph_ms, 4), round(dense_graph_ms, 4)
        row["speedup_vs_dense"] = round(dense_graph_ms / graph_ms, 3)
        row["speedup_basis"] = "hipGraph replay (device-bound); eager_speedup_vs_dense = module call vs torch op, host-bound"
    else:
        row["speedup_vs_dense"] = row["eager_speedup_vs_dense"]
        row["speedup_basis"] = "eager calls (graph capture unavailable)"
    if kernel_ms is not None:
        ex = planes * alg_flops / (kernel_ms * 1e-3) / 1e12
        row["roofline"] = {"bound": "mfma", "kernel_ms": round(kernel_ms, 4),
                           "achieved": round(ex, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(ex / PEAK_BF16_TFLOPS, 4),
                           "algorithmic_tflops": round(alg_flops / (kernel_ms * 1e-3) / 1e12, 1),
                           "executed_over_algorithmic": planes}
    return row


def run(device=None, iters: int = 50):
    dev = device or torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator().manual_seed(0)
    rows = []
    with torch.no_grad():
        # ---- TTLinearM: the four linear shapes of a DeiT-small block, 64 x 197 tokens
        hp = HPM.fresh_table("tt_deit_small_patch16_224_hp.HyperParamsDictRatio2x")
        T = 64 * 197
        for lname, fin, fout in (("blocks.1.attn.qkv.weight", 384, 1152), ("blocks.1.attn.proj.weight", 384, 384),
                                 ("blocks.1.mlp.fc1.weight", 384, 1536), ("blocks.1.mlp.fc2.weight", 1536, 384)):
            if lname not in hp.tt_shapes:
                continue
            lin = tt_layers.TTLinearM(fin, fout, bias=True, hp_dict=hp, name=lname).to(dev)
            # the dense baseline multiplies by the SAME weight (the cores' own product): GEMM time on this part moves by
            # several per cent with the operand values alone (clock / power), and two of the rows run the identical GEMM
            w_in0, w_out0 = lin._factors()
            w = (w_out0 @ w_in0).contiguous()
            b = lin.bias.detach().clone()
            rq = lin.tt_ranks[lin.out_tt_order]
            for dtype in (torch.float32, torch.bfloat16):
                x = torch.randn(64, 197, fin, generator=g).to(dev).to(dtype)
                wd, bd = w.to(dtype), b.to(dtype)
                ms, dense = _pair(_time, lambda: lin(x), lambda: F.linear(x, wd, bd), iters)
                gms, gdense = _pair(_graph_time, lambda: lin(x), lambda: F.linear(x, wd, bd), 20)
                kernel_ms, alg = None, 2.0 * T * rq * (fin + fout)
                if lin._fused_ok(x):
                    n = 1 if dtype == torch.bfloat16 else 3
                    w_in, w_out = lin._factors()
                    go = ops.chain_fused(x.reshape(-1, fin), HF.planes_of(w_in, n, pad_rows=64),
                                         HF.planes_of(w_out, n, pad_cols=64), lin.bias, fout, prepare_only=True)
                    kernel_ms = _time(go, iters)
                r = _row("TTLinearM deit_small %s (%d tokens, ranks %s)" % (lname[:-7], T, lin.tt_ranks), dtype, ms, dense,
                         kernel_ms, alg, 1 if kernel_ms is not None else lin.tt_order, gms, gdense)
                r["path"] = ("recovered dense weight on the library GEMM (TTLinearM._dense_pays: bf16 inference, chain saves "
                             "no arithmetic or little over a long reduction)" if lin._dense_pays(x, rq) else "fused chain kernel")
                r["chain_flops_per_token"] = int(sum(2 * c.numel() * _rest(lin, i) for i, c in enumerate(lin.tt_cores)))
                r["contracted_flops_per_token"] = int(2 * rq * (fin + fout))
                r["dense_flops_per_token"] = int(2 * fin * fout)
                rows.append(r)
        # ---- TTConv2dM: ResNet-18 layer3.1.conv1 (256, 256, 3, 3) at 14x14 and layer4.0.conv2 (512, 512, 3, 3) at 7x7, B = 64
        hp18 = HPM.fresh_table("tt_resnet18_hp.HyperParamsDictGeneralRatio2x")
        for lname, ch, hw in (("layer3.1.conv1.weight", 256, 14), ("layer4.0.conv2.weight", 512, 7)):
            conv = tt_layers.TTConv2dM(ch, ch, 3, padding=1, bias=False, hp_dict=hp18, name=lname).to(dev)
            wc = torch.randn(ch, ch, 3, 3, generator=g).to(dev)
            for dtype in (torch.float32, torch.bfloat16):
                xc = torch.randn(64, ch, hw, hw, generator=g).to(dev).to(dtype)
                wcd = wc.to(dtype)
                ms, dense = _pair(_time, lambda: conv(xc), lambda: F.conv2d(xc, wcd, None, 1, 1), iters)
                gms, gdense = _pair(_graph_time, lambda: conv(xc), lambda: F.conv2d(xc, wcd, None, 1, 1), 20)
                one = ops.conv_chain_pays(xc, conv.in_tt_ranks[0], conv.out_tt_ranks[-1], conv.kernel_size, conv.stride,
                                          conv.padding, conv.dilation)
                rows.append(_row("TTConv2dM resnet18 %s (B=64, %dx%d, ranks %s)" % (lname[:-7], hw, hw, conv.tt_ranks), dtype,
                                 ms, dense, None, 0.0, 1 if one else 3, gms, gdense))
        # ---- TKConv2dC: ResNet-32 layer3 3x3 (64, 64, 3, 3), x (128, 64, 8, 8)
        hp32 = HPM.fresh_table("tk_resnet32_hp.HyperParamsDictRatio3x")
        tk = tk_layers.TKConv2dC(64, 64, 3, padding=1, bias=False, hp_dict=hp32, name="layer3.1.conv1.weight").to(dev)
        wk = torch.randn(64, 64, 3, 3, generator=g).to(dev)
        for dtype in (torch.float32, torch.bfloat16):
            xk = torch.randn(128, 64, 8, 8, generator=g).to(dev).to(dtype)
            wkd = wk.to(dtype)
            ms, dense = _pair(_time, lambda: tk(xk), lambda: F.conv2d(xk, wkd, None, 1, 1), iters)
            gms, gdense = _pair(_graph_time, lambda: tk(xk), lambda: F.conv2d(xk, wkd, None, 1, 1), 20)
            one = ops.conv_chain_pays(xk, tk.in_rank, tk.out_rank, tk.kernel_size, tk.stride, tk.padding, tk.dilation)
            rows.append(_row("TKConv2dC resnet32 layer3.1.conv1 (B=128, 8x8, ranks [%d, %d])" % (tk.out_rank, tk.in_rank),
                             dtype, ms, dense, None, 0.0, 1 if one else 3, gms, gdense))
    return rows


def _rest(lin, i):
    """Number of times core i is applied per token in the per-core chain of TTLinear.py:79-86."""
    q = lin.out_tt_order
    if i >= q:                                  # input core i: applied once per combination of the earlier input modes
        rest = 1
        for j in range(q, i):
            rest *= lin.tt_shapes[j]
        return rest
    rest = 1                                     # output core i: once per combination of the later output modes
    for j in range(i + 1, q):
        rest *= lin.tt_shapes[j]
    return rest
