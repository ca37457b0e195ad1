"""Developer aid: sha1 of everything a few engine runs return (pose, correspondences, every stage tensor) -- run it with two builds
(RDM_LIB_PATH=<other .so>) to see whether a kernel change kept the bits.   python tools/bits_of_run.py"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdmnet_amd import config, weights, engine
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = config.make_cfg()
e = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
e.keep_taps(True)
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
sc = np.load(os.path.join(ROOT, 'tests', 'golden', 'scans.npz'))
h = hashlib.sha1()
for r, s in ((z['ref0'], z['src0']), (z['ref1'], z['src1']), (sc['s000000'], sc['s000004'])):
    res = e.run(torch.from_numpy(r).cuda(), torch.from_numpy(s).cuda())
    h.update(e.transform().tobytes())
    for c in e.corr():
        h.update(c.cpu().numpy().tobytes())
    for name in ('feats_c', 'nodes', 'matching_scores', 'ref_corr_points', 'estimated_transform'):
        h.update(e.tensor(name).cpu().numpy().tobytes())
    print(int(res.n_correspondences), h.hexdigest())
