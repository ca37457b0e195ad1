source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# driver-style short runs (--steps 20 --warmup 5: a 40 ms timed region) interleaved for the values of one knob:
#   bash tools/ab_short.sh VAR "v1 v2" [rounds]
K=$1; VALS=$2; N=${3:-4}
run() {
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-steps 0 --api-steps 0 --real-slots off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2), [round(x,2) for x in d['mean_ms_per_pair_by_quarter']])
"
}
for i in $(seq $N); do
  unset $K; run "default"
  for v in $VALS; do export $K=$v; run "$K=$v"; done
done
