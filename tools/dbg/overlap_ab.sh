# decoder on the side stream: A/B on one box (lab build knob RDM_DECODER_OVERLAP=0|1), one and four pairs in flight
cd "$GRAFT_REPO_ROOT"
export RDM_LIB_PATH=$PWD/rdmnet_amd/librdmnet_hip_lab.so
run() { python bench.py --streams $1 --steps 240 --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $1 overlap $2 ->', round(d['value'],1),'pairs/s p50',round(d['p50_ms_per_pair'],3), 'one pair', round(d['one_pair_in_flight']['p50_ms_per_pair'],3))
"; }
for i in 1 2; do
  for s in 1 4; do
    RDM_DECODER_OVERLAP=0 run $s 0
    RDM_DECODER_OVERLAP=1 run $s 1
  done
done
unset RDM_LIB_PATH
timeout 600 python -m pytest tests/test_engine_gpu.py tests/test_reference_goldens_gpu.py -x -q 2>&1 | tail -2
