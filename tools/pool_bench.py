"""Graph-replayed timings of the shortcut max-pool (gather_max, functional.py:54-67) of the four strided blocks on a bench pair's
real pyramid, in cell order as the engine runs it.   python tools/pool_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import _lib, config, engine, ops, weights
from kpconv_bench import timed

z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz'))
cfg = config.make_cfg()
eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
dd = eng.collate(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda())
g = torch.Generator().manual_seed(0)
print('| pool | M | H | C | real fill | us | GB/s (M H (8 + 4 C)) | of 8 TB/s |')
print('|---|---|---|---|---|---|---|---|')
for lvl, c in ((0, 64), (1, 128), (2, 256), (3, 512)):
    idx = dd['subsampling'][lvl]
    s = dd['points'][lvl]
    M, H = idx.shape
    x = ops.feat_empty(s.shape[0], c, 'cuda'); x.copy_(torch.randn(s.shape[0], c, generator=g))
    fill = float((idx < s.shape[0]).float().mean())
    want = torch.cat([x, torch.zeros(1, x.shape[1], device='cuda')])[idx.clamp(max=s.shape[0])].max(1)[0]
    got = ops.gather_max(x, idx)
    assert torch.equal(got, want), 'pool mismatch'
    t = timed(lambda: ops.gather_max(x, idx))
    nbytes = M * H * (8 + 4 * c)
    print(f'| L{lvl}->L{lvl + 1} | {M} | {H} | {c} | {fill:.2f} | {t:.1f} | {nbytes / t / 1e3:.0f} | {nbytes / t / 1e3 / 8000:.2f} |')
