#!/bin/bash
# A/B builds of the tile kernel (rdmnet_amd/librdmnet_hip_ab_*.so):  gpurun -- 'bash tools/r04_kp2.sh'
cd "$GRAFT_REPO_ROOT"
for lib in $(ls rdmnet_amd/librdmnet_hip_ab_*.so); do
  echo "== $lib"
  RDM_LIB_PATH=$PWD/$lib timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "lds_tile" 2>&1 | tail -1
  RDM_LIB_PATH=$PWD/$lib timeout 600 python tools/kpconv_bench.py 0 2>/dev/null | grep "cell order"
done
