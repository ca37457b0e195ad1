#!/bin/bash
# the driver's short run (--steps 20 --warmup 5) against the start-up stagger:  gpurun -- 'bash tools/dbg/short_sweep.sh'
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3 4; do
for s in 1.5 0 0.4 0.8 2.5; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --stagger-ms $s 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('stagger', $s, 'value', round(d['value'],1),'ms/step',round(d['ms_per_step'],3), 'roofline', round(d['roofline']['frac'],3))
"
done
done
