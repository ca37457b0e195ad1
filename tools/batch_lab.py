"""Lab for VERDICT r4 "next 3" (let one engine run B pairs as ONE launch sequence): what would stacking the rows of B pairs buy?

The engine takes two clouds.  Here the "pair" it is given is B bench pairs laid side by side -- pair b shifted by b x 400 m along
x, so that no neighbourhood, voxel or grid cell of one pair touches another's -- i.e. every level of the pyramid holds exactly
the rows a B-pair batch would stack, and every kernel whose work is linear in the rows (grid subsampling, radius searches,
KPConv, the encoder's / decoder's GEMMs, GroupNorm, pools) runs ONCE over B times the rows: the launch sequence of a batched
engine, without having to write one.  (The superpoint stages -- attention, NMS, grouping, coarse matching -- see B times the
tokens in ONE problem instead of B problems: their cost here is an over-estimate of a batched engine's, the per-class table of
tools/batch_lab.sh leaves them out of the comparison.)  GroupNorm statistics then span B pairs, so the RESULTS are not a
pair's: this is a cost model, not a product path.

  python tools/batch_lab.py B STREAMS [PAIRS]     -> pairs/s of `STREAMS` pipelines of B-stacked pairs
  bash tools/batch_lab.sh                          -> per-kernel-class time per ORIGINAL pair, B = 1 against B = 4 (rocprofv3)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from rdmnet_amd import config, pipeline, weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stacked(z, ids, spacing=400.0):
    refs, srcs = [], []
    for b, i in enumerate(ids):
        off = np.array([b * spacing, 0.0, 0.0], np.float32)
        refs.append(z[f'ref{i}'] + off)
        srcs.append(z[f'src{i}'] + off)
    return np.concatenate(refs), np.concatenate(srcs)


if __name__ == '__main__':
    B, streams = int(sys.argv[1]), int(sys.argv[2])
    n_stacked = int(sys.argv[3]) if len(sys.argv) > 3 else 48  # stacked pairs run in the timed part
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'synthetic_pairs.npz'))
    n_fix = len([k for k in z.files if k.startswith('ref')])
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    jobs = []
    for k in range(4):  # four distinct stacked inputs, cycled
        r, s = stacked(z, [(k + b) % n_fix for b in range(B)])
        jobs.append((torch.from_numpy(r).cuda(), torch.from_numpy(s).cuda()))
    pipe = pipeline.PairPipeline(cfg, state, pairs_in_flight=streams)

    def one(eng, i):
        eng.run(*jobs[i % len(jobs)])
        return True
    pipe.map(range(4 * streams), one)  # warm-up (arena growth for B x the points included)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.map(range(n_stacked), one)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f'B={B} streams={streams}: {n_stacked} stacked runs in {dt * 1e3:.1f} ms = {dt / n_stacked * 1e3:.3f} ms per stacked run, '
          f'{B * n_stacked / dt:.1f} original pairs/s')
