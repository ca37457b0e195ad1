source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"  # developer knobs live in the lab build
# HBM traffic (PMC) and per-layer times of the fused KPConv kernel (RDM_FUSED_KPCONV=1) beside the default gather + GEMM pair
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/profiles_r02; mkdir -p $O
db() { find "$1" -name "*.db" | head -1; }
for mode in fused default; do
  if [ $mode = fused ]; then export RDM_FUSED_KPCONV=1; else unset RDM_FUSED_KPCONV; fi
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $O/x_$c
    rocprofv3 --kernel-trace --pmc $c -d $O/x_$c -- python bench.py --steps 4 --warmup 2 --ramp-seconds 0 --no-cpu-baseline --streams 1 --host-steps 0 --api-steps 0 > $O/x.log 2>&1
  done
  python - "$(db $O/x_FETCH_SIZE)" "$(db $O/x_WRITE_SIZE)" $mode <<'PY' >> $O/r02_pmc_fused_kpconv.md
import sqlite3, sys
f, w, mode = sys.argv[1:4]
def per(db, counter):
    cur = sqlite3.connect(db).cursor()
    disp = {r[0]: (r[1], r[2]) for r in cur.execute('select name, count(*), sum(end-start) from kernels group by name')}
    vals = {r[0]: r[1] for r in cur.execute('select name, sum(counter_value) from pmc_events where counter_name=? group by name', (counter,))}
    pairs = max([v[0] for n, v in disp.items() if 'concat_points_kernel' in n] + [1])
    return disp, vals, pairs
df, vf, pairs = per(f, 'FETCH_SIZE')
dw, vw, _ = per(w, 'WRITE_SIZE')
print(f'\n### {mode}: per scan pair ({pairs} pairs in the run; counters in KB, FETCH_SIZE raw -- double it for wide reads, MI355X_MICROARCH.md §HBM)\n')
print('| kernel | launches/pair | us/pair | FETCH_SIZE MB/pair (raw) | WRITE_SIZE MB/pair |')
print('|---|---|---|---|---|')
tf = tw = tt = 0.0
for n in sorted(df, key=lambda n: -df[n][1]):
    if 'kpconv' not in n and not ('gemm_kernel' in n) and 'splitk' not in n:
        continue
    short = n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60]
    print(f'| `{short}` | {df[n][0] / pairs:.1f} | {df[n][1] / pairs / 1e3:.1f} | {vf.get(n, 0) / pairs / 1024:.1f} | {vw.get(n, 0) / pairs / 1024:.1f} |')
    tf += vf.get(n, 0) / pairs / 1024; tw += vw.get(n, 0) / pairs / 1024; tt += df[n][1] / pairs / 1e3
print(f'| **KPConv gather / fused + all gemm_kernel + split-K reduce** | | {tt:.1f} | {tf:.1f} | {tw:.1f} |')
PY
done
rm -rf $O/x_FETCH_SIZE $O/x_WRITE_SIZE
cat $O/r02_pmc_fused_kpconv.md
