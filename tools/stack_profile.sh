cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_stack; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/s4 -- python tools/stack_profile.py 4 10 > $O/s4.log 2>&1
python profiles/summarize_rocprof.py $(find $O/s4 -name "*.db" | head -1) 40 > $O/stack4.md
rm -rf $O/s4
head -50 $O/stack4.md
