"""Export the reference's rank tables (fixture G0) as JSON data.

Runs ONLY in the build container (needs /root/reference).  It imports every
``hp_dicts/*.py`` class of the reference and writes the numbers -- the data
contract -- into  dnn-compression-tensor-admm_amd/tadmm/data/hp_dicts.json.
No reference source text is copied, only ``ranks`` / ``tt_shapes`` values.
"""
import importlib, inspect, json, os, sys

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

out = {}
for fn in sorted(os.listdir(os.path.join(REF, "hp_dicts"))):
    if not fn.endswith("_hp.py"):
        continue
    modname = fn[:-3]
    mod = importlib.import_module("hp_dicts." + modname)
    for cname, cls in inspect.getmembers(mod, inspect.isclass):
        if cls.__module__ != mod.__name__:
            continue
        entry = {}
        for attr in ("ranks", "tt_shapes"):
            if hasattr(cls, attr):
                d = getattr(cls, attr)
                entry[attr] = {k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in d.items()}
                entry[attr + "_is_tuple"] = {k: isinstance(v, tuple) for k, v in d.items()}
        out[modname + "." + cname] = entry

dst = os.path.join(os.path.dirname(__file__), "..", "..", "dnn-compression-tensor-admm_amd", "tadmm", "data", "hp_dicts.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=0)
print(len(out), "tables ->", os.path.abspath(dst), os.path.getsize(dst), "bytes")
for k, v in out.items():
    print(k, len(v.get("ranks", {})), "tt" if "tt_shapes" in v else "")
