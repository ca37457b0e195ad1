source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# tools/exp_gemm_big.sh in the lock-step schedule: larger tiles (RDM_GEMM_BIG=<min rows>,<tile 4..7>: 128x64, 64x128, 128x128, 256x64)
# for the un-split products, and with RDM_GEMM_BIG_SPLIT=1 for the split-K products too (same split factor: same bits)
run() {
  python bench.py --steps 160 --warmup 8 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --layer-events-every 0 --real-slots off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$1 ->', round(d['value'],1),'pairs/s')
"
}
run default
for v in 500,6 500,4 500,5 1000,6 3000,6; do RDM_GEMM_BIG_SPLIT=1 RDM_GEMM_BIG=$v run "RDM_GEMM_BIG_SPLIT=1 RDM_GEMM_BIG=$v"; done
run default
