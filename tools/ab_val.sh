source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# interleaved comparison of the values of one environment knob on the same box:  bash tools/ab_val.sh VAR "v1 v2 ..." [rounds] [streams]
K=$1; VALS=$2; N=${3:-3}; S=${4:-4}
run() {
  python bench.py --streams $S --steps 320 --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2))
"
}
for i in $(seq $N); do
  unset $K; run "default"
  for v in $VALS; do export $K=$v; run "$K=$v"; done
done
