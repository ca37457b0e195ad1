python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > gpurun_out/t4.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
export RDM_BENCH_SHARE_DEVICE=1
for cfg in "2 4" "2 2" "3 3" "4 2" "4 1" "2 3"; do set -- $cfg
  python bench.py --gpus $1 --streams $2 --steps 240 --warmup 16 --ramp-seconds 3 --no-cpu-baseline --host-steps 0 --dist-backend gloo 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('procs $1 streams $2 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2))
" >> gpurun_out/exp1.log
done
cat gpurun_out/t4.log gpurun_out/smoke.log gpurun_out/exp1.log
