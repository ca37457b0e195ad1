"""How long does filling a pinned buffer take -- alone, and while another thread runs pairs on the GPU (spinning / polling waits)?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rdmnet_amd import config, engine, weights
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'synthetic_pairs.npz'))
ref, src = z['ref0'], z['src0']
pinned = torch.empty((40000, 3), dtype=torch.float32).pin_memory()
plain = torch.empty((40000, 3), dtype=torch.float32)
def fill(buf, n=200):
    t0 = time.perf_counter()
    for _ in range(n):
        buf[:ref.shape[0]] = torch.from_numpy(np.ascontiguousarray(ref, np.float32))
        buf[ref.shape[0]:ref.shape[0] + src.shape[0]] = torch.from_numpy(np.ascontiguousarray(src, np.float32))
    return (time.perf_counter() - t0) / n * 1e3
print('cpus', len(os.sched_getaffinity(0)), 'fill pinned %.3f ms, plain %.3f ms (GPU idle)' % (fill(pinned), fill(plain)))
cfg = config.make_cfg()
eng = engine.Engine(cfg, weights.synthetic_state_dict(cfg, seed=0))
r, s = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
for wait in (0, 50):
    eng.set_wait(wait)
    stop = threading.Event()
    lat = []
    def loop():
        with torch.cuda.stream(torch.cuda.Stream()):
            while not stop.is_set():
                t0 = time.perf_counter(); eng.run(r, s); lat.append((time.perf_counter() - t0) * 1e3)
    th = threading.Thread(target=loop); th.start()
    time.sleep(0.5)
    a = fill(pinned, 50); b = fill(plain, 50)
    n0 = len(lat)
    two = [threading.Thread(target=fill, args=(pinned, 50)) for _ in range(2)]
    [t.start() for t in two]; [t.join() for t in two]
    during = lat[n0:]
    stop.set(); th.join()
    print(f'wait {wait}: fill pinned {a:.3f} ms, plain {b:.3f} ms while an engine thread runs pairs; engine p50 {np.median(lat[:n0]):.2f} ms, with two filler threads {np.median(during) if during else -1:.2f} ms')
