# rocprofv3 kernel trace of one stream of lock-step groups (the GPU to one group at a time):  bash tools/trace_one_stream_groups.sh <tag> [ENV=..]
[ -n "$2" ] && source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-base}; shift
O=gpurun_out/trace_ls1_$TAG; rm -rf $O; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --stats -d $O/t -- python bench.py --steps 24 --warmup 4 --ramp-seconds 0 --no-cpu-baseline --streams 1 --lockstep 4 --host-steps 0 --api-steps 0 --full-steps 0 --real-slots off --layer-events-every 0 > $O/t.log 2>&1
python profiles/summarize_rocprof.py $(find $O/t -name "*.db" | head -1) 0 > $O/summary.md
