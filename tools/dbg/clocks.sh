#!/bin/bash
# shader clock and power while the bench runs (four pairs in flight, then one):  gpurun -- 'bash tools/dbg/clocks.sh'
cd "$GRAFT_REPO_ROOT"
for s in 4 1; do
python bench.py --steps 3000 --warmup 16 --ramp-seconds 2 --streams $s --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 > /tmp/b.json 2>/dev/null &
BP=$!
sleep 9
for i in 1 2 3 4 5; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power" | tr '\n' ' '; echo " [streams $s]"; sleep 0.5; done
wait $BP
python -c "
import json
for l in open('/tmp/b.json'):
    if l.startswith('{'): d=json.loads(l); print('streams $s value', round(d['value'],1))
"
done
echo idle; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|power"
