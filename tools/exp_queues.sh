for q in 4 5 6 8; do
GPU_MAX_HW_QUEUES=$q python bench.py --streams 4 --steps 360 --warmup 16 --ramp-seconds 3 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('GPU_MAX_HW_QUEUES=$q streams 4 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2))
"
done
