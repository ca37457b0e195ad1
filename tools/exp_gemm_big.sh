source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# larger GEMM tiles for the big un-split products with four pairs in flight (RDM_GEMM_BIG, gemm.hip)
run() {
  python bench.py --streams 4 --steps 240 --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('RDM_GEMM_BIG=${RDM_GEMM_BIG:-off} ->', round(d['value'],1),'pairs/s')
"
}
unset RDM_GEMM_BIG; run
for v in 3000,6 10000,6 3000,4 3000,5 1000,6; do export RDM_GEMM_BIG=$v; run; done
unset RDM_GEMM_BIG; run
