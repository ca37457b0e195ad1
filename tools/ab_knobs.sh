source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# Interleaved A/B of lab-knob settings in the lock-step schedule:  bash tools/ab_knobs.sh <rounds> "<VAR=..> <VAR=..>" "<..>" ...
# ("-" = no knob).  One short bench per setting and round; prints pairs/s.
N=$1; shift
run() {
  env $1 python bench.py --steps 160 --warmup 8 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --layer-events-every 0 --real-slots off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 ->', round(d['value'],1),'pairs/s', round(1e3 / d['value'],4), 'ms/pair')
"
}
for i in $(seq $N); do
  for s in "$@"; do
    if [ "$s" = "-" ]; then run "RDM_NOKNOB=1"; else run "$s"; fi
  done
done
