source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# quick throughput check: four pairs in flight and one (no CPU baseline, no host-to-host / API passes)
for s in 4 1 4; do
  python bench.py --streams $s --steps ${STEPS:-320} --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('streams $s ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2), 'gather frac', round(r['frac'],3), 'alone', round((r.get('one_pair_in_flight') or {}).get('frac',0),3), 'layer', round(r['kpconv_layer']['timed_region']['frac'],3), round((r['kpconv_layer'].get('one_pair_in_flight') or {}).get('frac',0),3))
"
done
