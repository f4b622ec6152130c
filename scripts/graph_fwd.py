"""hipGraph replay of the factorised-layer forwards against the dense op (GPU box): removes the host from the comparison."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
import torch.nn.functional as F
from tadmm import tt_layers, functional as HF, hp as HPM
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
hp = HPM.fresh_table("tt_deit_small_patch16_224_hp.HyperParamsDictRatio2x")

def graph_time(fn, iters=20, reps=50):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(iters): y = fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (iters * reps) * 1e3, y

with torch.no_grad():
    for lname, fin, fout in (("blocks.1.attn.proj.weight", 384, 384), ("blocks.1.mlp.fc2.weight", 1536, 384), ("blocks.1.mlp.fc1.weight", 384, 1536)):
        lin = tt_layers.TTLinearM(fin, fout, bias=True, hp_dict=hp, name=lname).to(dev)
        x = torch.randn(64, 197, fin, generator=g).to(dev).to(torch.bfloat16)
        wd = torch.randn(fout, fin, generator=g).to(dev).to(torch.bfloat16)
        bd = torch.randn(fout, generator=g).to(dev).to(torch.bfloat16)
        ref = lin(x).float()
        try:
            td, _ = graph_time(lambda: F.linear(x, wd, bd))
            tm, y = graph_time(lambda: lin(x))
            err = float((y.float() - ref).abs().max())
            print(lname, "graph replay us/call: dense %.2f | module %.2f | speedup %.3f | max diff vs eager %.3g" % (td, tm, td / tm, err), flush=True)
        except Exception as e:
            print(lname, "graph capture failed:", repr(e)[:300], flush=True)
