#!/bin/bash
# A/B of the batched collate (round 5):  gpurun -- 'bash tools/ab_collate_batch.sh'
cd "$GRAFT_REPO_ROOT"
run() {
  python bench.py "$@" --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$*: %.1f pairs/s  p50 %.2f ms  frac %.3f' % (d['value'], d['p50_ms_per_pair'], r['frac']))"
}
for rep in 1 2; do
  for b in 1 2 4 5 8; do run --steps 20 --warmup 5 --collate-batch $b; done
done
for b in 1 4 8; do run --collate-batch $b; done
