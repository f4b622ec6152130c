"""TTConv2dM at the ResNet-18 ImageNet stages (B = 64): module call vs dense conv2d, fp32 and bf16."""
import os, sys, json
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
import torch
import torch.nn.functional as F
from tadmm import hp as HPM, tt_layers
from tadmm.fwdbench import _time
dev = torch.device("cuda", 0)
hp18 = HPM.fresh_table("tt_resnet18_hp.HyperParamsDictGeneralRatio2x")
g = torch.Generator().manual_seed(0)
with torch.no_grad():
    for name, cin, cout, hw in (("layer1.0.conv1.weight", 64, 64, 56), ("layer2.1.conv1.weight", 128, 128, 28),
                                ("layer3.1.conv1.weight", 256, 256, 14), ("layer4.1.conv1.weight", 512, 512, 7)):
        conv = tt_layers.TTConv2dM(cin, cout, 3, padding=1, bias=False, hp_dict=hp18, name=name).to(dev)
        w = torch.randn(cout, cin, 3, 3, generator=g).to(dev)
        for dtype in (torch.float32, torch.bfloat16):
            x = torch.randn(64, cin, hw, hw, generator=g).to(dev).to(dtype)
            wd = w.to(dtype)
            ms = _time(lambda: conv(x), 20)
            dense = _time(lambda: F.conv2d(x, wd, None, 1, 1), 20)
            print(json.dumps({"layer": name, "hw": hw, "ranks": conv.tt_ranks, "dtype": str(dtype)[6:], "ms": round(ms, 4),
                              "dense_ms": round(dense, 4), "speedup": round(dense / ms, 2)}), flush=True)
