source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
for c in 256 128 96 64 256 128; do
RDM_GEMM_CUS=$c python bench.py --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('RDM_GEMM_CUS=$c', round(d['value'],1), 'p50', round(d['p50_ms_per_pair'],2))
"
done
