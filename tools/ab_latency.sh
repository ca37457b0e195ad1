source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# Interleaved A/B of lab-knob settings on ONE pair in flight:  bash tools/ab_latency.sh <rounds> "<VAR=..>" ...   ("-" = no knob)
# prints the latency-mode p50 and the serial p50 (ms per pair) of bench.py's one-stream passes.
N=$1; shift
run() {
  env $1 python bench.py --steps 24 --warmup 4 --streams 1 --ramp-seconds 1 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --layer-events-every 0 --real-slots off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); o=d['one_pair_in_flight']; print('$1 -> latency mode p50', round(o['p50_ms_per_pair'],3), 'ms, serial p50', round(o['serial_p50_ms_per_pair'],3), 'ms, timed', round(d['value'],1), 'pairs/s')
"
}
for i in $(seq $N); do
  for s in "$@"; do
    if [ "$s" = "-" ]; then run "RDM_NOKNOB=1"; else run "$s"; fi
  done
done
