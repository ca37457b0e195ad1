for cfg in "5 1" "6 2" "8 4" "6 3" "4 2" "4 4"; do set -- $cfg
RDM_BENCH_HIGH_PRIORITY_STREAMS=$2 python bench.py --streams $1 --steps 240 --warmup 16 --ramp-seconds 3 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $1 (high priority: $2) ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2))
"
done
