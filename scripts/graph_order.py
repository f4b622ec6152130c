"""Order effects of hipGraph replay timing: dense F.linear and the TTLinearM dense-path module, alternating (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
import torch.nn.functional as F
from tadmm import tt_layers, fwdbench, hp as HPM
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
hp = HPM.fresh_table("tt_deit_small_patch16_224_hp.HyperParamsDictRatio2x")
with torch.no_grad():
    for lname, fin, fout in (("blocks.1.attn.proj.weight", 384, 384), ("blocks.1.mlp.fc2.weight", 1536, 384)):
        lin = tt_layers.TTLinearM(fin, fout, bias=True, hp_dict=hp, name=lname).to(dev)
        w_in0, w_out0 = lin._factors()
        wd = (w_out0 @ w_in0).contiguous().to(torch.bfloat16); bd = lin.bias.detach().clone().to(torch.bfloat16)
        x = torch.randn(64, 197, fin, generator=g).to(dev).to(torch.bfloat16)
        lin(x)
        c = lin.__dict__["_chain_cache"]
        print(lname, "same weight bits:", torch.equal(c["dense"], wd), "same bias:", torch.equal(c["dense_bias"], bd),
              "ptr align", c["dense"].data_ptr() % 256, wd.data_ptr() % 256, x.data_ptr() % 256)
        for rnd in range(3):
            a = fwdbench._graph_time(lambda: F.linear(x, wd, bd), 20)
            b = fwdbench._graph_time(lambda: lin(x), 20)
            d = fwdbench._graph_time(lambda: F.linear(x, c["dense"], c["dense_bias"]), 20)
            print("   round", rnd, "dense %.2f us | module %.2f us | F.linear on the cached tensors %.2f us" % (a * 1e3, b * 1e3, d * 1e3), flush=True)
