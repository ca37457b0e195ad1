import os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from rdmnet_amd import config, weights, engine
cfg = config.make_cfg()
e0 = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
es = [e0] + [engine.Engine(cfg, None, share_with=e0) for _ in range(3)]
z = np.load('/root/repo/tests/golden/synthetic_pairs.npz')
pairs = [(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda()), (torch.from_numpy(z['ref1']).cuda(), torch.from_numpy(z['src1']).cuda())] * 2
r = e0.run(*pairs[0]); print('plain run arena_used MiB', r.arena_used / 2**20)
e0.keep_taps(True); r = e0.run(*pairs[0]); print('keep_taps run arena_used MiB', r.arena_used / 2**20); e0.keep_taps(False)
with torch.cuda.stream(torch.cuda.Stream()):
    res = engine.Engine.run_lockstep(es, pairs)
    print('lockstep collated: arena_used MiB per engine', [x.arena_used / 2**20 for x in res])
    res = engine.Engine.run_lockstep(es, pairs, collate_batched=False)
    print('lockstep not collated:', [x.arena_used / 2**20 for x in res])
