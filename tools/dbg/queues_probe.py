"""What makes one pair slower after a four-pairs-in-flight region in the same process?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from rdmnet_amd import config, engine, weights, synthetic, pipeline
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
host_pairs = synthetic.cached_pairs(8, os.path.join(ROOT, 'gpurun_out', 'bench_pairs'), os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
pairs = [(torch.from_numpy(r).cuda(), torch.from_numpy(s_).cuda()) for r, s_, _ in host_pairs]
cfg = config.make_cfg()
state = weights.synthetic_state_dict(cfg, seed=0)
eng = engine.Engine(cfg, state)
st = torch.cuda.Stream()

def measure(tag, e=eng, s=st, n=120):
    lat = []
    with torch.cuda.stream(s):
        for i in range(n + 16):
            t0 = time.perf_counter()
            e.run(*pairs[i % 8])
            if i >= 16:
                lat.append((time.perf_counter() - t0) * 1e3)
    print(f'{tag}: p50 {np.median(lat):.3f} ms', flush=True)

measure('fresh')
eng.set_pairs_in_flight(4); eng.set_pairs_in_flight(1)
measure('after toggling the hint')
eng.set_pairs_in_flight(4)
with torch.cuda.stream(st):
    for i in range(40):
        eng.run(*pairs[i % 8])
eng.set_pairs_in_flight(1)
measure('after 40 runs with the hint at 4 (one stream)')
mode = sys.argv[1] if len(sys.argv) > 1 else 'pipe'
if mode == 'pipe':
    pipe = pipeline.PairPipeline(cfg, state, pairs_in_flight=4, engines=[eng])
    t0 = time.perf_counter()
    out = pipe.run_pairs([pairs[i % 8] for i in range(400)])
    print('pipeline: %.1f pairs/s' % (400 / (time.perf_counter() - t0)))
    for e in pipe.engines:
        e.set_pairs_in_flight(1)
    measure('after the four-in-flight region, first engine, own stream')
    measure('... first engine, the pipeline\'s first stream', eng, pipe.streams[0])
    measure('... second engine', pipe.engines[1], st)
    fresh = engine.Engine(cfg, None, share_with=eng)
    measure('... a fresh engine', fresh, torch.cuda.Stream())
else:
    import threading
    def idle():
        with torch.cuda.stream(torch.cuda.Stream()):
            torch.zeros(16, device='cuda').add_(1)
            torch.cuda.current_stream().synchronize()
    ts = [threading.Thread(target=idle) for _ in range(4)]
    [t.start() for t in ts]; [t.join() for t in ts]
    measure('after four threads touched the runtime')
