"""Where does the harness spend its time with ONE pair in flight?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rdmnet_amd import dataset as ds, synthetic, config, weights, infer, evaluation
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fixture = os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz')
base = ds.ArrayPairDataset(synthetic.cached_pairs(8, os.path.join(ROOT, 'gpurun_out', 'bench_pairs'), fixture))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
data = ds.CyclingPairDataset(base, 128)
cfg = config.make_cfg()
cfg.neighbor_limits = [65, 63, 69, 70, 81]
t = infer.Tester(cfg, weights.synthetic_state_dict(cfg, seed=0), save_npz=False, pairs_in_flight=n)
acc = {'run': 0.0, 'measure': 0.0, 'n': 0}
orig = t._work
def timed(eng, job):
    item, r, s = job
    t0 = time.perf_counter(); res = eng.run(r.contiguous(), s.contiguous()); t1 = time.perf_counter()
    T = eng.transform(); rc, sc, cs = eng.host_corr()
    m = t.summary.measure(np.asarray(item['transform'], np.float64), T, rc, sc, cs); t2 = time.perf_counter()
    acc['run'] += t1 - t0; acc['measure'] += t2 - t1; acc['n'] += 1
    return {'seq_id': item['seq_id'], 'ref_frame': item['ref_frame'], 'src_frame': item['src_frame'], 'n_corr': int(res.n_correspondences), 'ms': 0, 'transform': T, '_metrics': m}
t._work = timed
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 2 * n
workers = int(sys.argv[3]) if len(sys.argv) > 3 else max(2, n)
stager = ds.PairStager(data, depth=depth, workers=workers)
import collections, traceback
tally = collections.Counter()
stop_s = threading.Event()
def sampler():
    me = threading.get_ident()
    names = {}
    while not stop_s.is_set():
        time.sleep(0.005)
        names = {th.ident: th.name for th in threading.enumerate()}
        for tid, fr in sys._current_frames().items():
            if tid == me:
                continue
            st = traceback.extract_stack(fr)[-3:]
            tally[(names.get(tid, '?'), ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}:{f.name}' for f in reversed(st)))] += 1
if os.environ.get('SAMPLE'):
    threading.Thread(target=sampler, daemon=True).start()
t0 = time.perf_counter()
t.run(stager)
stop_s.set()
dt = time.perf_counter() - t0
print(f'pairs in flight {n} depth {depth} workers {workers}: {128 / dt:.1f} pairs/s; per pair: engine run {acc["run"] / acc["n"] * 1e3:.2f} ms, transform + host_corr + measure {acc["measure"] / acc["n"] * 1e3:.2f} ms; pipeline stats {t.pipeline.last_stats}')

for (name, where), c in tally.most_common(14):
    print(f'{c:5d} {name:12s} {where}')
