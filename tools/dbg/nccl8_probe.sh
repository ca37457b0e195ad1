cd "$GRAFT_REPO_ROOT"
RDM_BENCH_SHARE_DEVICE=1 timeout 300 python bench.py --gpus 8 --steps 16 --warmup 4 --ramp-seconds 0 --pairs 2 --host-steps 0 --api-steps 0 --full-steps 0 --no-cpu-baseline --real-slots off --dist-backend nccl 2>&1 | grep -iE "duplicate|invalid usage|ncclInvalid|Error|error" | sort | uniq -c | sort -rn | head -8
