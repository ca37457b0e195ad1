import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from rdmnet_amd import _lib, ops, synthetic
ref, src, _ = synthetic.make_pair(0)
pts = torch.from_numpy(np.concatenate([ref, src])).cuda()
lens = torch.tensor([len(ref), len(src)], dtype=torch.int64).cuda()
L = _lib.lib()
L.rdm_debug_gs_profile.restype = ctypes.c_int
L.rdm_debug_gs_profile.argtypes = [ctypes.c_void_p]
v = 0.6
for lvl in range(4):
    for _ in range(3):
        cap, l2 = ops.grid_subsample_device(pts, lens, v)
    torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 32)()
    L.rdm_debug_gs_profile(ctypes.addressof(buf))
    t = np.array(buf[:7], dtype=np.float64) / 100.0  # 100 MHz -> us
    print('level', lvl, 'n', pts.shape[0], 'phases us (P0,P1,P2,P3-6,P7,P8):', np.round(np.diff(t), 1), 'total', round(t[6] - t[0], 1))
    pts = cap[:int(l2.sum())].contiguous(); lens = l2; v *= 2
