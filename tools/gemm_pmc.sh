# Lab: rocprofv3 --pmc passes over tools/gemm_pmc_probe.py (decoder3 at four pairs' rows: 15516 x 1536 x 512), per kernel:
# duration, matrix-pipe busy, waits, LDS activity.   bash tools/gemm_pmc.sh <tag> [ENV=..]
source "$(dirname "${BASH_SOURCE[0]}")/lab_env.sh"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/gemm_pmc_$TAG; rm -rf $O; mkdir -p $O
env "$@" rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/a -- python tools/gemm_pmc_probe.py > $O/a.log 2>&1
env "$@" rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM -d $O/b -- python tools/gemm_pmc_probe.py > $O/b.log 2>&1
python - "$O" <<'P'
import sqlite3, sys, glob, re
for sub in ('a', 'b'):
    db = glob.glob(f'{sys.argv[1]}/{sub}/**/*.db', recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    disp = {r[0]: (r[1], r[2]) for r in cur.execute('select name, count(*), sum(end-start) from kernels group by name').fetchall()}
    k = {}
    for name, c, v in cur.execute('select name, counter_name, sum(counter_value) from pmc_events group by name, counter_name').fetchall():
        k.setdefault(name, {})[c] = v
    for name in sorted(k, key=lambda n: -disp.get(n, (0, 0))[1])[:3]:
        calls, ns = disp[name]
        short = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', name.replace('(anonymous namespace)::', '').replace('void ', ''))[:60]
        print(f'{short}: {calls} calls, {ns / calls / 1e3:.1f} us each; per call: ' + ', '.join(f'{c} {v / calls / 1e6:.2f}M' for c, v in sorted(k[name].items())))
P
