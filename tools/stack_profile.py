"""Kernel-time distribution when every launch is fat enough to fill the GPU: B scenes side by side in one pair of clouds
(tools/scale_probe.py's stand-in), one stream.  Run under rocprofv3 --kernel-trace --stats."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdmnet_amd import config, engine, weights
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
cfg = config.make_cfg()
state = weights.synthetic_state_dict(cfg, seed=0)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'synthetic_pairs.npz'))
ref = np.concatenate([z[f'ref{i}'] + np.array([300.0 * i, 0, 0], np.float32) for i in range(B)])
src = np.concatenate([z[f'src{i}'] + np.array([300.0 * i, 0, 0], np.float32) for i in range(B)])
r, s = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
eng = engine.Engine(cfg, state)
for _ in range(runs):
    eng.run(r, s)
torch.cuda.synchronize()
