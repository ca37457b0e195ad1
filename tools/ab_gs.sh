source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for v in multi single multi single; do
  if [ $v = single ]; then export RDM_GS_SINGLE=1; else unset RDM_GS_SINGLE; fi
  python bench.py --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v', round(d['value'],1), 'p50', round(d['p50_ms_per_pair'],2))
"
done
for v in multi single; do
  if [ $v = single ]; then export RDM_GS_SINGLE=1; else unset RDM_GS_SINGLE; fi
  python bench.py --streams 1 --steps 160 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v streams 1', round(d['value'],1), 'p50', round(d['p50_ms_per_pair'],2))
"
done
