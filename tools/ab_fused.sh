source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
python -m pytest tests/test_ops_gpu.py tests/test_reference_goldens_gpu.py tests/test_engine_gpu.py tests/test_encoder_gpu.py -q -m gpu 2>&1 | tail -30 > gpurun_out/t9.log; grep -E "^E  |passed|failed|^FAILED" gpurun_out/t9.log | head -20
for v in 0 1; do
  if [ $v = 0 ]; then export RDM_FUSED_KPCONV=1; else unset RDM_FUSED_KPCONV; fi
  python bench.py --no-cpu-baseline --host-steps 0 --api-steps 0 > gpurun_out/b4_$v.json 2> gpurun_out/b4_$v.err
  cp gpurun_out/bench_layers.json gpurun_out/layers_$v.json
done
