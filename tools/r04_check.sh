#!/bin/bash
# round-4 sanity run on the GPU box:  gpurun -- 'bash tools/r04_check.sh'
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_check
mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
for l in open('gpurun_out/r04_check/bench.json'):
    if l.startswith('{'):
        d=json.loads(l)
        print('value',round(d['value'],1),'p50',round(d['p50_ms_per_pair'],2),'iso',d['one_pair_in_flight'],'full',d['full_tables'] and round(d['full_tables']['value'],1),'h2h',round(d['host_to_host']['value'],1),'api',round(d['drop_in_api']['value'],1))
        r=d['roofline']; print('roofline',round(r['frac'],3),'iso',round(r['one_pair_in_flight']['frac'],3), 'cpu', d['cpu_baseline'] and d['cpu_baseline']['value'])
PY
timeout 600 python -m rdmnet_amd.infer --synthetic 512 --no-npz --neighbor-limits 65 63 69 70 81 > $O/infer.log 2>&1; echo "infer rc=$?"; tail -4 $O/infer.log
