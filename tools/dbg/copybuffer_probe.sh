#!/bin/bash
# Where do the __amd_rocclr_copyBuffer dispatches of a four-stream bench run come from?  Kernel traces with
# and without the per-layer HIP events, one and four streams; prints the blit count per pair and the kernels right before them.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for cfg in "4 8" "4 0" "1 8" "1 0"; do
  set -- $cfg
  rm -rf gpurun_out/cb
  rocprofv3 --kernel-trace -d gpurun_out/cb -- python bench.py --steps 16 --warmup 4 --ramp-seconds 0 --no-cpu-baseline --host-steps 0 --api-steps 0 --real-slots off --streams $1 --layer-events-every $2 > gpurun_out/cb.log 2>&1
  python - "$1" "$2" $(find gpurun_out/cb -name "*.db" | head -1) <<'P'
import sqlite3, sys, collections
streams, every, db = sys.argv[1:]
cur = sqlite3.connect(db).cursor()
rows = cur.execute('select name, start, end, stream_id, queue_id, grid_x from kernels order by start').fetchall()
pairs = sum(1 for r in rows if 'nms_kernel' in r[0])
blits = [i for i, r in enumerate(rows) if 'copyBuffer' in r[0]]
prev = collections.Counter()
for i in blits:
    s = rows[i][3]
    j = i - 1
    while j >= 0 and rows[j][3] != s:
        j -= 1
    prev[rows[j][0].split('(')[0][-40:] if j >= 0 else 'first'] += 1
print(f'streams {streams} layer-events-every {every}: {pairs} pairs, {len(blits)} copyBuffer dispatches = {len(blits) / max(pairs, 1):.1f} per pair; grids {collections.Counter(rows[i][5] for i in blits).most_common(3)}; '
      f'preceded on their stream by {prev.most_common(4)}')
P
done
rm -rf gpurun_out/cb
