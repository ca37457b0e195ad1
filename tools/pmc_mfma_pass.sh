cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r02; R=r02; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/pm -- python bench.py --steps 4 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 --host-steps 0 --api-steps 0 > $O/pm.log 2>&1
DB=$(find $O/pm -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -- python bench.py --steps 4 --warmup 2 --streams 1   (MI355X, $R, one pair in flight)"; echo; python profiles/summarize_mfma.py $DB; } > $O/${R}_pmc_mfma.md
rm -rf $O/pm
head -12 $O/${R}_pmc_mfma.md
