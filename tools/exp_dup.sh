source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# Marginal cost of each kernel class with four pairs in flight: RDM_DUP=<class> launches every kernel of the class twice
# (idempotent launches, same results), so  ms/pair(dup) - ms/pair(base)  is what the class costs per pair in the
# throughput regime -- the number that says where a faster kernel would pay (sums of kernel durations do not: small
# kernels overlap with other pairs' work).  One pair in flight for comparison.
run() {
  python bench.py --streams $1 --steps 240 --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 --full-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $1 dup ${RDM_DUP:-none} ->', round(d['value'],1),'pairs/s', round(d['ms_per_step'],4), 'ms/pair')
"
}
for s in 4 1; do
  unset RDM_DUP; run $s
  for c in gemm gemmsmall fused gather gnapply gnfin pool attn tail sinkhorn rn gs splitk; do
    export RDM_DUP=$c; run $s
  done
  unset RDM_DUP; run $s
done
