#!/bin/bash
# Regenerates the round's profile summaries on the GPU box:  gpurun -- 'bash profiles/collect.sh r01'
# Kernel traces and PMC counters are collected in separate rocprofv3 runs (see MI355X_MICROARCH.md).
set -u
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_$R
mkdir -p $O
db() { find "$1" -name "*.db" | head -1; }

rocprofv3 --kernel-trace --stats -d $O/t4 -- python bench.py --steps 20 --warmup 4 --ramp-seconds 0 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off > $O/t4.log 2>&1
{ printf '%s\n' "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --ramp-seconds 0 --no-cpu-baseline   (MI355X, $R, default: 4 streams x lock-step groups of 4 pairs -- 'grouped <body>' rows are rdm::grouped_kernel launches serving the pairs of a group, steps = pairs; durations include cross-stream sharing; 24 timed + warm-up steps (4 pairs each) and the one-stream passes after them -- 6 groups + 8 single pairs with per-layer events, 16 serial, 28 in the engine's latency mode; the __amd_rocclr_copyBuffer rows are the parameter uploads of the first engine's construction (417; the other engines share its parameters), before any pair runs)"; echo; python profiles/summarize_rocprof.py $(db $O/t4) 0; } > $O/${R}_kernel_trace_default_4streams.md
rocprofv3 --kernel-trace --stats -d $O/t1 -- python bench.py --steps 20 --warmup 4 --ramp-seconds 0 --no-cpu-baseline --streams 1 --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off > $O/t1.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 20 --warmup 4 --ramp-seconds 0 --no-cpu-baseline --streams 1   (MI355X, $R, one pair in flight, i.e. the engine in its latency mode -- the first level's search and blocks run beside the subsampling chain and the decoder beside the second transformer on a side stream, so some durations include that sharing; 24 steps + the passes after them: 8 pairs with per-layer events and 16 more with the mode off, 28 with it on; the __amd_rocclr_copyBuffer rows are the parameter uploads of the first engine's construction (417; the other engines share its parameters), before any pair runs)"; echo; python profiles/summarize_rocprof.py $(db $O/t1) 0; } > $O/${R}_kernel_trace_stream1.md
cp gpurun_out/bench_layers.json $O/${R}_kpconv_layers_events.json
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -- python bench.py --steps 4 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 --lockstep 4 --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off > $O/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -- python bench.py --steps 4 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 --lockstep 4 --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off > $O/pw.log 2>&1
SRC="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bench.py --steps 4 --warmup 2 --streams 1 --lockstep 4 (one stream, lock-step groups of 4 pairs: 2 warm-up + 4 timed + 6 roofline-pass groups, then 44 one-pair runs of the latency passes), MI355X, $R"
{ echo "# $SRC"; echo "# raw counter values in KB; FETCH_SIZE must be doubled for wide coalesced reads on gfx950 (MI355X_MICROARCH.md §HBM)"; echo; python profiles/summarize_pmc.py $(db $O/pf) $(db $O/pw) 0; } > $O/${R}_pmc_fetch_write.md
python profiles/pmc_gather_json.py $(db $O/pf) $(db $O/pw) "$SRC" grouped > $O/${R}_pmc_kpconv_gather.json
python profiles/pmc_gather_json.py $(db $O/pf) $(db $O/pw) "$SRC" > $O/${R}_pmc_kpconv_gather_one_pair.json
# the marginal cost of every kernel class in the lock-step schedule (lab build, RDM_DUP): txt + the json bench.py reads (roofline.timed_marginal)
bash tools/exp_dup_lockstep.sh > $O/${R}_marginal_cost.body 2>/dev/null
{ echo "# tools/exp_dup_lockstep.sh on the lab build (librdmnet_hip_lab.so), MI355X, $R, lock-step schedule (4 streams x groups of 4 pairs, bench.py --steps 160): RDM_DUP=<class> issues every launch of the class twice; ms/pair(dup) - ms/pair(none) = the class's marginal cost"; cat $O/${R}_marginal_cost.body; } > $O/${R}_marginal_cost.txt
python profiles/marginal_json.py $O/${R}_marginal_cost.txt "tools/exp_dup_lockstep.sh, lab build, MI355X, $R (profiles/${R}_marginal_cost.txt)" > $O/${R}_marginal_cost.json
rm -f $O/${R}_marginal_cost.body
# matrix-core utilisation (north star: "MFMA utilisation against gfx950 peaks"): one SQ pass, one pair in flight
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $O/pm -- python bench.py --steps 4 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off > $O/pm.log 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE -- python bench.py --steps 4 --warmup 2 --streams 1   (MI355X, $R, one pair in flight)"; echo; python profiles/summarize_mfma.py $(db $O/pm); } > $O/${R}_pmc_mfma.md
rm -rf $O/t4 $O/t1 $O/pf $O/pw $O/pm
ls -la $O
