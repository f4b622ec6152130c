#!/bin/bash
# usage (GPU box): bash scripts/phases_ab.sh <config> "<VAR=val ...>" ...  -- ms_per_step, instrumented phases and filter block per setting
CFG=$1; shift
for s in "$@"; do
  env $s python bench.py --config $CFG --no-cpu-baseline --no-forward --no-per-layer 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('== $s', round(d['ms_per_step'],3), {k:round(v,3) for k,v in d['phases_ms'].items()})
print('   filter', d.get('filter'))
ro=d.get('roofline_other',{})
for k in ('second_kernel','eig_f64'):
    v=ro.get(k)
    if v: print('   ',k,{a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a not in ('note','jacobi_tick3')})
"
done
