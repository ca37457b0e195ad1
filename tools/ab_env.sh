source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# interleaved A/B of one environment knob on the same box:  bash tools/ab_env.sh RDM_NO_VIRTUAL_CONCAT [rounds] [streams]
K=$1; N=${2:-3}; S=${3:-4}
run() {
  python bench.py --streams $S --steps 320 --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2))
"
}
for i in $(seq $N); do
  unset $K; run "default      "
  export $K=1; run "$K=1"
done
