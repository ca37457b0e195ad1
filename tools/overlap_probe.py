"""One pair in flight: latency of rdm_engine_run with and without the latency mode (rdm_engine_set_overlap), runs taking turns,
and the check that both give the same bits.   python tools/overlap_probe.py [pairs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import config, engine, weights

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz'))
n_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfg = config.make_cfg()
eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
from rdmnet_amd import synthetic
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
host_pairs = synthetic.cached_pairs(8, os.path.join(ROOT, 'gpurun_out', 'bench_pairs'), os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
pairs = [(torch.from_numpy(r).cuda(), torch.from_numpy(s_).cuda()) for r, s_, _ in host_pairs]  # the bench workload's 8 pairs
stream = torch.cuda.Stream()


configs = [0, 2]
lat = {c: [] for c in configs}
outs = {}
engs = {c: (eng if k == 0 else engine.Engine(cfg, None, share_with=eng)) for k, c in enumerate(configs)}  # one engine (and side stream) per configuration
with torch.cuda.stream(stream):
    for c in configs:  # builds the side streams, warms up
        engs[c].set_overlap(c)
        for _ in range(10):
            engs[c].run(*pairs[0])
    for i in range(n_pairs):  # the configurations take turns pair by pair: clock drift hits all of them alike
        for c in configs:
            t0 = time.perf_counter()
            engs[c].run(*pairs[i % len(pairs)])
            lat[c].append((time.perf_counter() - t0) * 1e3)
            if i == 0:
                outs[c] = (engs[c].transform().copy(), [a.copy() for a in engs[c].host_corr()])
base = outs[configs[0]]
for c in configs:
    a = np.array(lat[c])
    same = np.array_equal(outs[c][0], base[0]) and all(np.array_equal(x, y) for x, y in zip(outs[c][1], base[1]))
    print('  per pair id (ms):', ' '.join(f'{np.median(a[k::len(pairs)]):.2f}' for k in range(len(pairs))), ' points:', ' '.join(str(r.shape[0] + s_.shape[0]) for r, s_ in pairs) if c == 0 else '')
    print(f'overlap mode {c}: p50 {np.median(a):.3f} ms  mean {a.mean():.3f}  p10 {np.percentile(a, 10):.3f}  same bits: {same}')
