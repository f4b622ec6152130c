"""Golden fixture G7: the full selection ladder of the reference's ``utils.get_hp_dict``
(utils.py:258-400) -- outcome of every (model_name, ratio, format, tt_type) combination of a grid that
covers each rung, each prefix form and the failure modes.

Runs ONLY in the build container (imports /root/reference/utils.py).  Recorded per combination: the
table the reference returns ('<hp file stem>.<class name>'), 'None', 'Exception:<message>' or
'ImportError' (the ladder names a few classes its hp files never define).  Data only, no source text.
"""
import itertools, json, os, sys

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
import utils as ref_utils  # noqa: E402

MODELS = ["deit_tiny_patch16_224", "deit_small_patch16_224", "resnet32", "resnet56", "resnet18", "resnet50",
          "mobilenetv2", "mobilenetv2_cifar", "densenet40", "densenet121", "densenet201", "vgg16", "vgg16_bn",
          "unknown_net"]
PREFIXES = ["", "tk_", "tt_", "svd_", "tkc_", "tkm_", "tkr_", "ttm_", "ttr_", "svdc_"]
RATIOS = ["1.5", "2", "3", "4", "5", "10", "sc", "7"]
FORMATS = ["none", "tk", "tt", "svd"]
TT_TYPES = ["general", "special"]

out = {}
for m, pre, ratio, fmt, ttt in itertools.product(MODELS, PREFIXES, RATIOS, FORMATS, TT_TYPES):
    if pre and fmt != "none":          # a prefix overrides the format: one format value is enough
        continue
    key = f"{pre}{m}|{ratio}|{fmt}|{ttt}"
    try:
        r = ref_utils.get_hp_dict(pre + m, ratio, fmt, ttt)
        out[key] = "None" if r is None else f"{r.__module__.split('.')[-1]}.{r.__name__}"
    except ImportError:
        out[key] = "ImportError"
    except Exception as e:             # the reference raises bare Exception
        out[key] = "Exception:" + str(e)

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "g7_hp_ladder.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=0, sort_keys=True)
vals = {}
for v in out.values():
    vals[v.split(":")[0] if v.startswith("Exception") else ("table" if "." in v else v)] = vals.get(v.split(":")[0] if v.startswith("Exception") else ("table" if "." in v else v), 0) + 1
print(len(out), "combinations ->", dst, vals)
