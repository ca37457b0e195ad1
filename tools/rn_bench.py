"""Times rdm_radius_neighbors (count-only vs full) on a level-0-like self search (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdmnet_amd import _lib, ops
from rdmnet_amd import synthetic
ref, src, _ = synthetic.make_pair(0)  # the bench workload: 2 x ~16 k points
pts = torch.from_numpy(np.concatenate([ref, src])).cuda()
lens = torch.tensor([len(ref), len(src)], dtype=torch.int64).cuda()
L = _lib.lib()
n = pts.shape[0]
ws = torch.empty(L.rdm_radius_neighbors_workspace_bytes(n, n, 2), dtype=torch.uint8, device='cuda')
flags = torch.zeros(2, dtype=torch.int32, device='cuda')
out = torch.empty((n, 65), dtype=torch.int64, device='cuda')
def run(width, radius=1.275):
    _lib.check(L.rdm_radius_neighbors(pts.data_ptr(), n, pts.data_ptr(), n, lens.data_ptr(), lens.data_ptr(), 2, radius, width,
                                      out.data_ptr() if width else 0, 0, flags.data_ptr(), flags[1:].data_ptr(), ws.data_ptr(), ws.numel(),
                                      _lib.stream_ptr()), 'rn')
for width in (0, 65):
    for _ in range(3): run(width)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run(width)
    e1.record(); torch.cuda.synchronize()
    print('width', width, 'us per call', e0.elapsed_time(e1) * 100, 'max count', int(flags[0]))
