"""Runs one GPU test function repeatedly in ONE process and reports every failure (race / uninitialised-read hunting).
usage: python scripts/repeat_test.py tests/test_gpu_chain.py test_ttlinearm_fused_backward_matches_fp64_autograd 200"""
import importlib.util, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "dnn-compression-tensor-admm_amd"))
path, name, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
spec = importlib.util.spec_from_file_location("t", os.path.join(ROOT, path))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
fn = getattr(mod, name)
import torch
fails = 0
for i in range(n):
    try:
        fn()
    except AssertionError:
        fails += 1
        print("FAIL at repetition", i)
        traceback.print_exc(limit=2)
    if i % 7 == 3:     # perturb the allocator state between repetitions
        junk = [torch.empty(1 + (i * 37) % 5000, device="cuda") for _ in range(5)]
        del junk
print("%d failures in %d repetitions" % (fails, n))
