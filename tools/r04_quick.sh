#!/bin/bash
# short bench + harness rate:  gpurun -- 'bash tools/r04_quick.sh'
cd "$GRAFT_REPO_ROOT"
python bench.py --steps 320 --warmup 16 --ramp-seconds 3 --no-cpu-baseline --host-steps 0 --api-steps 96 --full-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('value', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2), 'one pair', round(d['one_pair_in_flight']['p50_ms_per_pair'],3), 'api', round(d['drop_in_api']['value'],1), 'roofline', round(d['roofline']['frac'],3), 'alone', round(d['roofline']['one_pair_in_flight']['frac'],3))
"
python -m rdmnet_amd.infer --synthetic 512 --no-npz --neighbor-limits 65 63 69 70 81 2>&1 | tail -6
