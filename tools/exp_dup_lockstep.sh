source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# tools/exp_dup.sh for the lock-step schedule (round 5: 4 streams x groups of 4 pairs, the bench's default): RDM_DUP=<class> issues
# every launch of the class twice (idempotent), so ms/pair(dup) - ms/pair(none) is what the class costs per pair in this regime.
run() {
  python bench.py --steps 160 --warmup 8 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 --layer-events-every 0 --real-slots off 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('lockstep 4x4 dup ${RDM_DUP:-none} ->', round(d['value'],1),'pairs/s', round(1e3 / d['value'],4), 'ms/pair')
"
}
unset RDM_DUP; run
for c in gemm gemmsmall fused gather gnapply gnfin pool attn tail sinkhorn rn gs splitk lgr p2n coarse rnbuild rows ln ups; do
  export RDM_DUP=$c; run
done
unset RDM_DUP; run
