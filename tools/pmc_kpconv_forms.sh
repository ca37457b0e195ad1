#!/bin/bash
# Counter comparison of the two one-kernel KPConv forms, per layer shape (round 4):  gpurun -- 'bash tools/pmc_kpconv_forms.sh'
# Separate rocprofv3 --pmc passes (no trace domains besides --kernel-trace), one process per shape.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_kpconv_forms; rm -rf $O; mkdir -p $O
PASSES=("TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_REQ_sum TCC_HIT_sum" "FETCH_SIZE" "WRITE_SIZE"
        "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY")
for shape in 0 1 2 3; do
  for i in "${!PASSES[@]}"; do
    rocprofv3 --kernel-trace --pmc ${PASSES[$i]} -d $O/s${shape}_p$i -- python tools/kpconv_forms_once.py $shape 5 > $O/s${shape}_p$i.log 2>&1 || tail -3 $O/s${shape}_p$i.log
  done
done
python tools/pmc_kpconv_forms_summary.py $O > $O/r04_pmc_kpconv_forms.md
cat $O/r04_pmc_kpconv_forms.md
find $O -name "*.db" -delete; find $O -type d -empty -delete
