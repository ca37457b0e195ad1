cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest tests/test_engine_gpu.py tests/test_pipeline_gpu.py tests/test_native_gpu.py -x -q 2>&1 | tail -3
bash tools/r04_quick.sh 2>&1 | head -3
