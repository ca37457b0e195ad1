#!/bin/bash
# KPConv tile kernel: parity tests + graph-replayed per-layer timings + phase stamps:  gpurun -- 'bash tools/r04_kp.sh'
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04_kp
mkdir -p $O
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "kpconv" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 600 python tools/kpconv_bench.py 0 2> $O/kpconv_bench.err | grep -v "gather alone" > $O/kpconv_bench.md; echo "bench rc=$?"; cat $O/kpconv_bench.md
RDM_LIB_PATH=$PWD/rdmnet_amd/librdmnet_hip_timing.so timeout 300 python tools/tile_lab.py 2>&1 | grep -v amdgpu.ids | grep -A1 "cell order" > $O/tile_lab.txt; cat $O/tile_lab.txt
