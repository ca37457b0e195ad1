"""Phases of the drop-in operator path (collate + model(data_dict)) per pair, four pairs in flight."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rdmnet_amd import collate, config, model, pipeline, synthetic, weights
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cfg = config.make_cfg()
cfg.neighbor_limits = [65, 63, 69, 70, 81]
state = weights.synthetic_state_dict(cfg, seed=0)
dev = torch.device('cuda', 0)
pairs = synthetic.cached_pairs(8, os.path.join(ROOT, 'gpurun_out', 'bench_pairs'), os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
dev_pairs = [(torch.from_numpy(r).to(dev), torch.from_numpy(s).to(dev)) for r, s, _ in pairs]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
pipe = pipeline.PairPipeline(cfg, state, device=dev, pairs_in_flight=n)
net = model.create_model(cfg).cuda()
net.load_state_dict(state)
net.pairs_in_flight = n
acc = {'ones': 0.0, 'collate': 0.0, 'forward': 0.0, 'cpu': 0.0, 'n': 0}
import threading
lock = threading.Lock()
def step(eng, i):
    r, s_ = dev_pairs[i % 8]
    t0 = time.perf_counter()
    item = {'ref_points': r, 'src_points': s_, 'ref_feats': torch.ones((r.shape[0], 1), device=dev), 'src_feats': torch.ones((s_.shape[0], 1), device=dev)}
    t1 = time.perf_counter()
    data = collate.registration_collate_fn_stack_mode([item], cfg.backbone.num_stages, cfg.backbone.init_voxel_size, cfg.backbone.init_radius,
                                                      cfg.neighbor_limits, device=dev, engine=net.engine())
    data['testing'] = True
    t2 = time.perf_counter()
    out = net(data)
    t3 = time.perf_counter()
    out['estimated_transform'].cpu()
    t4 = time.perf_counter()
    with lock:
        acc['ones'] += t1 - t0; acc['collate'] += t2 - t1; acc['forward'] += t3 - t2; acc['cpu'] += t4 - t3; acc['n'] += 1
pipe.map(range(8 * n), step)
for k in acc: acc[k] = 0
t0 = time.perf_counter()
pipe.map(range(256), step)
dt = time.perf_counter() - t0
print(f'{n} in flight: {256 / dt:.1f} pairs/s; per pair (ms): ' + ', '.join(f'{k} {acc[k] / acc["n"] * 1e3:.3f}' for k in ('ones', 'collate', 'forward', 'cpu')))
def eng_step(eng, i):
    eng.run(*dev_pairs[i % 8])
pipe.map(range(8 * n), eng_step)
t0 = time.perf_counter(); pipe.map(range(256), eng_step); print(f'engine.run alone: {256 / (time.perf_counter() - t0):.1f} pairs/s')
