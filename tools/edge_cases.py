import numpy as np, torch, sys
sys.path.insert(0, '.')
from rdmnet_amd import config, engine, weights
cfg = config.make_cfg()
eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
rng = np.random.default_rng(0)
def cloud(n, scale):
    return torch.from_numpy((rng.uniform(-1, 1, (n, 3)) * np.array([scale, scale, 2.0])).astype(np.float32)).cuda()
for n, sc in [(3000, 30.0), (500, 20.0), (200, 10.0), (50, 5.0), (5, 1.0), (1, 1.0), (20000, 2.0), (4000, 200.0)]:
    try:
        r = eng.run(cloud(n, sc), cloud(max(n - 3, 1), sc))
        print(n, sc, 'ok: levels', list(r.level_sizes), 'nodes', r.n_ref_nodes, r.n_src_nodes, 'corr', r.n_correspondences, 'finite', bool(np.isfinite(eng.transform()).all()))
    except RuntimeError as e:
        print(n, sc, 'RuntimeError:', str(e)[:150])
