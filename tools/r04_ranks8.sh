#!/bin/bash
# Eight ranks on ONE device (the launcher / sharding / gather path of an 8-GPU node, minus the other seven GPUs):
#   gpurun -- 'bash tools/r04_ranks8.sh'
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_ranks8; mkdir -p $O
nproc; python - <<'PY'
import os
print('affinity', len(os.sched_getaffinity(0)), 'cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip() if os.path.exists('/sys/fs/cgroup/cpu.max') else None)
PY
run() {  # name, backend, extra args...
  name=$1; backend=$2; shift 2
  /usr/bin/time -v -o $O/$name.time env RDM_BENCH_SHARE_DEVICE=1 timeout 600 python bench.py --gpus 8 --steps 64 --warmup 8 --ramp-seconds 2 --pairs 4 \
      --host-steps 0 --api-steps 0 --full-steps 0 --no-cpu-baseline --real-slots off --dist-backend $backend "$@" > $O/$name.json 2> $O/$name.err
  echo "$name rc=$?"; grep -E "Elapsed|Maximum resident|Percent of CPU|User time|System time" $O/$name.time | tr '\n' ';'; echo
  tail -2 $O/$name.err | cut -c1-300
  python - $O/$name.json <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print({k: d[k] for k in ('value', 'n_gpus', 'steps', 'p50_ms_per_pair')}, d['records'], d['collective'], d['config']['wait'], d['config']['host_cpus_per_rank'], d['config'].get('host_cpus_pinned'))
PY
}
run nccl8 nccl
run gloo8 gloo
run gloo8_nopin gloo --pin off
run gloo8_spin gloo --wait-us 0
