"""How does the engine's time per point change with cloud size?  (Stacking B pairs into one pyramid multiplies the rows
of every encoder launch by B; a single pair of B-times larger clouds is a cheap stand-in for that.)"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdmnet_amd import config, engine, synthetic, weights
cfg = config.make_cfg()
state = weights.synthetic_state_dict(cfg, seed=0)
pairs = [synthetic.make_pair(i) for i in range(4)]
for B in (1, 2, 4):
    # B scenes side by side (300 m apart) in one pair of clouds: every encoder launch sees B times the rows
    ref = np.concatenate([pairs[i][0] + np.array([300.0 * i, 0, 0], np.float32) for i in range(B)])
    src = np.concatenate([pairs[i][1] + np.array([300.0 * i, 0, 0], np.float32) for i in range(B)])
    r, s = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
    for streams in (1, 4):
        engs = [engine.Engine(cfg, state) for _ in range(streams)]
        sts = [torch.cuda.Stream() for _ in range(streams)]
        def work(k, n):
            with torch.cuda.stream(sts[k]):
                for _ in range(n):
                    engs[k].run(r, s)
        for k in range(streams): work(k, 6)
        torch.cuda.synchronize()
        n = 60
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(k, n)) for k in range(streams)]
        [t.start() for t in ts]; [t.join() for t in ts]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (n * streams)
        print(f'{B} scene(s) per cloud, {len(ref)} points/scan: {streams} in flight: {dt * 1e3:.2f} ms/run = {dt * 1e3 / B:.2f} ms per scene', flush=True)
        del engs
