for cfg in resnet50_tt resnet18_tt deit_small_tt; do
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --config $cfg 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$cfg', round(d['ms_per_step'],2),'ms', round(d['value'],2),'it/s', 'svdGF/s',round(d['svd_gflops_per_s'],1), {k:round(v,2) for k,v in d['phases_ms'].items()})
print('   gram frac', round(d['roofline']['frac'],3), 'gemm TF', round(d['roofline_other']['gemm_f32_mfma']['achieved_tflops'],1), 'hbm GB/s', round(d['roofline_other']['hbm_sweeps']['achieved_gbs']), 'cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))"
done
