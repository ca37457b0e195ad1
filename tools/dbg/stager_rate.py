"""How fast does dataset.PairStager deliver pairs on its own, and through a PairPipeline whose job does nothing?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rdmnet_amd import dataset as ds, synthetic
fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'synthetic_pairs.npz')
base = ds.ArrayPairDataset(synthetic.cached_pairs(2, 'gpurun_out/bench_pairs', fixture))
data = ds.CyclingPairDataset(base, 512)
torch.cuda.init()
for depth, workers in ((2, 2), (8, 4)):
    t0 = time.perf_counter()
    n = 0
    for item, r, s in ds.PairStager(data, depth=depth, workers=workers):
        n += 1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'stager alone depth={depth} workers={workers}: {n / dt:.0f} pairs/s')
t0 = time.perf_counter()
for i in range(512):
    data[i]
print(f'dataset items alone: {512 / (time.perf_counter() - t0):.0f} items/s')
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for item, r, s in ds.PairStager(data, depth=8, workers=4):
    pass
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
