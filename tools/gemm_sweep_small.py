"""Tile choice for the skinny-K unary products at levels 0-1 (developer knob RDM_GEMM_TUNE); timed through rocprof-free
HIP events, 40 repetitions, operands resident."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from rdmnet_amd import ops
shapes = [(32000, 64, 32), (32000, 32, 128), (32000, 64, 128), (32000, 128, 32), (10961, 128, 64), (10961, 64, 256),
          (10961, 128, 256), (10961, 256, 64), (3879, 256, 128), (3879, 128, 512), (3879, 256, 512), (32000, 16, 64)]
for m, k, n in shapes:
    a = torch.randn(m, k, device='cuda'); b = torch.randn(k, n, device='cuda'); rd = torch.ones(m, device='cuda')
    res = []
    for tile in (0, 1, 2, 3):
        os.environ['RDM_GEMM_TUNE'] = f'{tile},0'
        for _ in range(3):
            ops.gemm(a, b, k, n, rowdiv=rd)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(40):
            ops.gemm(a, b, k, n, rowdiv=rd)
        e1.record(); torch.cuda.synchronize()
        res.append((tile, e0.elapsed_time(e1) / 40 * 1e3))
    print(f'M={m} K={k} N={n}: ' + ', '.join(f't{t}: {us:.1f}us' for t, us in res), flush=True)
