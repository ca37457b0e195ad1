"""Lab: pairs/s of STREAMS host threads, each driving a lock-step group of B engines on its stream (rdm_engine_run_lockstep), against
the product's schedule (B = 1: rdm_engine_run per pair).   python tools/lockstep_lab.py B STREAMS [PAIRS]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import config, engine, synthetic, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B, streams = int(sys.argv[1]), int(sys.argv[2])
n_pairs = int(sys.argv[3]) if len(sys.argv) > 3 else 192
cfg = config.make_cfg()
if os.environ.get('LS_SINKHORN_ITERS'):  # (experiment: how does the schedule respond to REMOVED work?)
    cfg.model.num_sinkhorn_iterations = int(os.environ['LS_SINKHORN_ITERS'])
state = weights.synthetic_state_dict(cfg, seed=0)
pairs = synthetic.cached_pairs(8, os.path.join(ROOT, 'gpurun_out', 'bench_pairs'), os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(s).cuda()) for r, s, _ in pairs]
first = engine.Engine(cfg, state)
groups = [[first if (g == 0 and k == 0) else engine.Engine(cfg, None, share_with=first) for k in range(B)] for g in range(streams)]
for grp in groups:
    for e in grp:
        e.set_pairs_in_flight(int(os.environ.get('LS_PIF', streams * B if streams * B >= 3 else streams)))
pool = [torch.cuda.Stream() for _ in range(16)]  # (consecutive streams of torch's pool)
pick = [int(x) for x in os.environ['LS_STREAMS'].split(',')] if os.environ.get('LS_STREAMS') else list(range(streams))
lock = threading.Lock()
state_ = {'next': 0, 'limit': 0}


def worker(g, stagger):
    torch.set_num_threads(1)
    with torch.cuda.stream(pool[pick[g]]):
        time.sleep(stagger)
        while True:
            with lock:
                i = state_['next']
                if i >= state_['limit']:
                    return
                take = min(B, state_['limit'] - i)
                state_['next'] = i + take
            batch = [dev[(i + k) % len(dev)] for k in range(take)]
            if B == 1:
                groups[g][0].run(*batch[0])
            else:
                engine.Engine.run_lockstep(groups[g], batch, collate_batched=os.environ.get('LS_COLLATE', '1') != '0')
            for e in groups[g][:take]:
                e.host_corr()


def run(n):
    state_['next'], state_['limit'] = 0, n
    ts = [threading.Thread(target=worker, args=(g, 0.0015 * g)) for g in range(streams)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for _ in range(3):
    run(8 * streams * B)
import ctypes
L = first.L
st = (ctypes.c_longlong * 8)()
try:
    L.rdm_lockstep_stats(st, 1)
except AttributeError:
    L = None
dt = run(n_pairs)
print(f'B={B} streams={streams} (pool {pick[:streams]}): {n_pairs} pairs in {dt * 1e3:.1f} ms = {n_pairs / dt:.1f} pairs/s')
if L is not None and B > 1:
    L.rdm_lockstep_stats(st, 0)
    runs = max(st[5], 1)
    print(f'  per group run: {st[0] / runs / 1e6:.2f} ms on the host thread, {st[1] / runs / 1e6:.2f} ms of it in {st[2] / runs:.1f} waits; '
          f'{st[3] / runs:.0f} grouped launches carrying {st[4] / runs:.0f} records')
if L is not None and os.environ.get('RDM_LOCKSTEP_STATS'):
    sys.stdout.flush()
    L.rdm_lockstep_stats_dump()
