for cfg in "4 interleave" "4 blocks" "8 interleave" "8 blocks" "6 blocks" "2 blocks"; do set -- $cfg
RDM_BENCH_CU_MASK=$2 python bench.py --streams $1 --steps 240 --warmup 16 --ramp-seconds 3 --no-cpu-baseline --host-steps 0 --api-steps 0 2>gpurun_out/cu.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $1 CU mask $2 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2))
"
tail -1 gpurun_out/cu.err | cut -c1-200
done
