#!/bin/bash
# kernel timelines of one pair in flight, serial against latency mode:  gpurun -- 'bash tools/timeline.sh'  -> gpurun_out/tl/timeline_m{0,2}.txt
# (under rocprofv3 the host's launches are several times slower, so the streams' relative positions are only indicative)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tl
for m in 0 2; do
rocprofv3 --kernel-trace -d gpurun_out/tl/m$m -- python tools/one_pair.py $m 12 > gpurun_out/tl/m$m.log 2>&1
tail -1 gpurun_out/tl/m$m.log
python tools/timeline.py $(find gpurun_out/tl/m$m -name "*.db" | head -1) 12 > gpurun_out/tl/timeline_m$m.txt
rm -rf gpurun_out/tl/m$m
done
