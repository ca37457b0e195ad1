source tools/lab_env.sh
python tools/gemm_wide_check.py gpurun_out/gw_off.json > gpurun_out/gw_off.txt 2>&1
RDM_GEMM_WIDE=1 python tools/gemm_wide_check.py gpurun_out/gw_on.json > gpurun_out/gw_on.txt 2>&1
python - <<'P'
import json
a=json.load(open('gpurun_out/gw_off.json')); b=json.load(open('gpurun_out/gw_on.json'))
for x,y in zip(a,b):
    same = all(x[k]==y[k] for k in ('sha_x1','sha_x4'))
    print(f"{x['name']:28s} bits {'same' if same else 'DIFF'}  x1 {x['us_x1']:7.1f} -> {y['us_x1']:7.1f} us ({y['plan_x1'][:2]})  x4 {x['us_x4']:7.1f} -> {y['us_x4']:7.1f} us ({y['tf_x4']:.0f} TF)")
P
tail -1 gpurun_out/gw_off.txt; tail -1 gpurun_out/gw_on.txt
