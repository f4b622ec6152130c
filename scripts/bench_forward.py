"""Forward throughput of the factorised layers: TTLinearM (DeiT-small qkv),
TTConv2dM (ResNet-18 layer4 3x3) and TKConv2dC (ResNet-32 layer3 3x3) against the dense layer they replace.
python scripts/bench_forward.py   -> one line per layer: ms, algorithmic GFLOP, TFLOP/s, dense ms (rocBLAS/MIOpen fp32)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
import torch.nn.functional as F
from tadmm import hp as HPM, tt_layers, tk_layers

dev = torch.device("cuda", 0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def main():
    g = torch.Generator().manual_seed(0)
    # --- TTLinearM: DeiT-small blocks.1.attn.qkv (1152 x 384), T = 64 * 197 tokens
    hp = HPM.fresh_table("tt_deit_small_patch16_224_hp.HyperParamsDictRatio2x")
    name = "blocks.1.attn.qkv.weight"
    lin = tt_layers.TTLinearM(384, 1152, bias=True, hp_dict=hp, name=name).to(dev)
    x = torch.randn(64, 197, 384, generator=g).to(dev)
    T = 64 * 197
    shapes, ranks = list(hp.tt_shapes[name]), list(hp.ranks[name])
    # chain FLOPs per token: sum over cores of 2 * r_i * n_i * r_{i+1} * (rest of the token's modes)
    w = torch.randn(1152, 384, generator=g).to(dev)
    with torch.no_grad():
        ms = timeit(lambda: lin(x))
        msd = timeit(lambda: F.linear(x, w))
    fl = 2.0 * T * sum(p.numel() for p in lin.tt_cores)      # lower bound; the chain's true count is below
    print("TTLinearM  deit_small qkv   T=%d: %.3f ms   dense F.linear fp32 %.3f ms   (dense %.1f GFLOP)" %
          (T, ms, msd, 2.0 * T * 1152 * 384 / 1e9))
    # --- TTConv2dM: ResNet-18 layer4.0.conv2 (512,512,3,3), x (64,512,7,7)
    hp18 = HPM.fresh_table("tt_resnet18_hp.HyperParamsDictGeneralRatio2x")
    n18 = "layer4.0.conv2.weight"
    conv = tt_layers.TTConv2dM(512, 512, 3, padding=1, bias=False, hp_dict=hp18, name=n18).to(dev)
    xc = torch.randn(64, 512, 7, 7, generator=g).to(dev)
    wc = torch.randn(512, 512, 3, 3, generator=g).to(dev)
    with torch.no_grad():
        ms = timeit(lambda: conv(xc))
        msd = timeit(lambda: F.conv2d(xc, wc, None, 1, 1))
    print("TTConv2dM  resnet18 layer4.0.conv2 B=64: %.3f ms   dense conv2d fp32 %.3f ms" % (ms, msd))
    # --- TKConv2dC: ResNet-32 layer3 3x3 (64,64,3,3) ranks from the 3x table, x (128,64,8,8)
    hp32 = HPM.fresh_table("tk_resnet32_hp.HyperParamsDictRatio3x")
    n32 = "layer3.1.conv1.weight"
    tk = tk_layers.TKConv2dC(64, 64, 3, padding=1, bias=False, hp_dict=hp32, name=n32).to(dev)
    xk = torch.randn(128, 64, 8, 8, generator=g).to(dev)
    wk = torch.randn(64, 64, 3, 3, generator=g).to(dev)
    with torch.no_grad():
        ms = timeit(lambda: tk(xk))
        msd = timeit(lambda: F.conv2d(xk, wk, None, 1, 1))
    print("TKConv2dC  resnet32 %s B=128: %.3f ms   dense conv2d fp32 %.3f ms" % (n32, ms, msd))
    # --- bf16 inference path of TTLinearM (last: the bf16 library calls of the dense baseline change MIOpen's
    # algorithm choice for the convolutions above when they run first)
    with torch.no_grad():
        xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
        msb = timeit(lambda: lin(xb))
        msdb = timeit(lambda: F.linear(xb, wb))
    print("TTLinearM  deit_small qkv   T=%d bf16: %.3f ms   dense F.linear bf16 %.3f ms" % (T, msb, msdb))


if __name__ == "__main__":
    main()
