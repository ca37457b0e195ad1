source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# second half of tools/exp_dup.sh: the smaller kernel classes (what is left of the 2.1 ms per pair after the big ones)
run() {
  python bench.py --streams $1 --steps 240 --warmup 16 --ramp-seconds 2 --no-cpu-baseline --host-steps 0 --api-steps 0 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('streams $1 dup ${RDM_DUP:-none} ->', round(d['value'],1),'pairs/s', round(d['ms_per_step'],4), 'ms/pair')
"
}
for s in 4; do
  unset RDM_DUP; run $s
  for c in splitk lgr p2n coarse rnbuild rows ln ups; do
    export RDM_DUP=$c; run $s
  done
  unset RDM_DUP; run $s
done
