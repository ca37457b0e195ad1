source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# Which form of the fine-level KPConv layers pays in the lock-step schedule (4 streams x groups of 4)?  All forms give the same bits.
run() {
  python bench.py --steps 160 --warmup 8 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --layer-events-every 0 --real-slots off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 ->', round(d['value'],1),'pairs/s')
"
}
for rep in 1 2; do
  run default
  RDM_KPCONV_TILE=0 run "RDM_KPCONV_TILE=0 (c_in 32/64 on kpconv_fused_kernel)"
  RDM_FUSED_KPCONV=0 run "RDM_FUSED_KPCONV=0 (gather + GEMM on every layer)"
done
