#!/bin/bash
# pairs-in-flight sweep of the bench value:  gpurun -- 'bash tools/dbg/streams_sweep.sh'
cd "$GRAFT_REPO_ROOT"
for s in 4 3 5 6 8 4; do
python bench.py --steps 320 --warmup 16 --ramp-seconds 3 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --streams $s 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams', $s, 'value', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],2), 'roofline', round(d['roofline']['frac'],3))
"
done
