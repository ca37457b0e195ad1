import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rdmnet_amd import config, weights, engine
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = config.make_cfg()
state = weights.synthetic_state_dict(cfg, seed=0)
scans = np.load(os.path.join(ROOT, 'tests', 'golden', 'scans.npz'))
a, b, c = scans['s000000'], scans['s000004'], scans['s000007']
crop = lambda p, r: np.ascontiguousarray(p[np.linalg.norm(p[:, :2], axis=1) < r])
distinct = [(crop(a, 10.0), crop(b, 10.0)), (crop(a, 16.0), crop(b, 16.0)), (crop(a, 12.0), crop(c, 12.0)), (crop(b, 9.0), crop(a, 9.0)), (crop(a, 20.0), crop(b, 20.0))]
dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(s).cuda()) for r, s in distinct]
e0 = engine.Engine(cfg, state)
engs = [e0] + [engine.Engine(cfg, None, share_with=e0) for _ in range(3)]
want = []
for r, s in dev:
    res = e0.run(r, s)
    want.append((e0.transform().copy(), int(res.n_correspondences)))
    print('serial', r.shape[0], s.shape[0], int(res.n_correspondences), [int(x) for x in res.level_sizes])
with torch.cuda.stream(torch.cuda.Stream()):
    for collated in (False, True):
        for combo in ([0, 1, 2, 3], [4, 4, 0, 3], [1, 2, 0, 0], [4, 2, 3, 1], [3, 3], [0, 3], [3, 0], [2, 3, 4]):
            try:
                res = engine.Engine.run_lockstep(engs, [dev[i] for i in combo], collate_batched=collated)
                ok = [np.array_equal(engs[k].transform(), want[i][0]) and int(res[k].n_correspondences) == want[i][1] for k, i in enumerate(combo)]
                print(collated, combo, ok)
            except RuntimeError as ex:
                print(collated, combo, 'ERROR', str(ex)[-80:])
