#!/bin/bash
source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# gpurun -- 'bash tools/pmc_gemm.sh M K N'  : MFMA / wait / LDS counters of the tiled GEMM on one shape
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_WAVES"; do
  rm -rf gpurun_out/pmc1
  rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc1 -- python tools/gemm_one.py $1 $2 $3 10 rowdiv > gpurun_out/pmc.log 2>&1
  python tools/pmc_dump.py $(find gpurun_out/pmc1 -name "*.db" | head -1) gemm_kernel
done
rm -rf gpurun_out/pmc1
rocprofv3 --kernel-trace --stats -d gpurun_out/pmc1 -- python tools/gemm_one.py $1 $2 $3 10 rowdiv > gpurun_out/pmc.log 2>&1
python profiles/summarize_rocprof.py $(find gpurun_out/pmc1 -name "*.db" | head -1) 1 | grep -E "gemm|splitk"
rm -rf gpurun_out/pmc1
