source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# Lock-step schedule: 16-deep k-tiles for the 64x64 products (17 KB of LDS, ~70 registers: twice the resident workgroups; same bits)?
run() {
  python bench.py --steps 160 --warmup 8 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --layer-events-every 0 --real-slots off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 ->', round(d['value'],1),'pairs/s')
"
}
for rep in 1 2; do run default; RDM_GEMM_BK16=1 run RDM_GEMM_BK16=1; done
