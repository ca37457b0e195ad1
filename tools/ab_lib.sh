# interleaved A/B of two builds of the library (same ABI) on one box:  bash tools/ab_lib.sh <other librdmnet_hip.so> [rounds] [streams]
OTHER=$1; N=${2:-3}; S=${3:-4}
run() {
  python bench.py --streams $S --steps 320 --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$1 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2), 'gather alone', round((r.get('group_alone') or r.get('single_pair_alone') or {}).get('frac',0),3))
"
}
for i in $(seq $N); do
  unset RDM_LIB_PATH; run "this build "
  export RDM_LIB_PATH=$OTHER; run "other build"
done
