#!/bin/bash
# Driver-style short run (20 timed pairs) against the start-up stagger of the pipeline's workers:  gpurun -- 'bash tools/exp_stagger.sh'
cd "$GRAFT_REPO_ROOT"
for rep in 1 2; do
for st in 1.5 0.75 0.4 0.0; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off --stagger-ms $st 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('stagger $st: %.1f pairs/s  p50 %.2f ms  quarters %s' % (d['value'], d['p50_ms_per_pair'], [round(x,2) for x in d['mean_ms_per_pair_by_quarter']]))"
done; done
