"""N runs of rdm_engine_run on one synthetic pair, one pair in flight, on a stream of its own (for rocprofv3 timelines).
python tools/one_pair.py OVERLAP_MODE [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import config, engine, weights

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz'))
mode = int(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
cfg = config.make_cfg()
eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
eng.set_overlap(mode)
ref, src = torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda()
lat = []
with torch.cuda.stream(torch.cuda.Stream()):
    for i in range(n):
        t0 = time.perf_counter()
        eng.run(ref, src)
        lat.append((time.perf_counter() - t0) * 1e3)
print('overlap mode', mode, 'p50 %.3f ms' % np.median(lat[n // 2:]))
