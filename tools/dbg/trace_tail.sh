cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace_tail; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/t -- python bench.py --steps 8 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off --layer-events-every 0 > $O/t.log 2>&1
python profiles/summarize_rocprof.py $(find $O/t -name "*.db" | head -1) 0 | grep -i "tail\|total kernel" 
