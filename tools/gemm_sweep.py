"""Sweeps tile shape x split-K factor of rdm_gemm for KPConv-shaped products (developer knob RDM_GEMM_TUNE).
python tools/gemm_sweep.py  (on the GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from rdmnet_amd import ops

shapes = [(3879, 1920, 128), (5289, 1920, 128), (1310, 3840, 256), (1902, 3840, 256), (563, 7680, 512), (819, 7680, 512),
          (10961, 960, 64), (3879, 960, 64), (1310, 1920, 128), (563, 3840, 256), (819, 2048, 512), (819, 512, 2048),
          (1902, 1284, 1024), (5289, 1536, 512)]
for m, k, n in shapes:
    a = torch.randn(m, k, device='cuda'); b = torch.randn(k, n, device='cuda'); rd = torch.ones(m, device='cuda')
    res = []
    for tile in (1, 2):
        for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24, 32):
            if sp > k // 64:
                continue
            os.environ['RDM_GEMM_TUNE'] = f'{tile},{sp}'
            try:
                for _ in range(3):
                    ops.gemm(a, b, k, n, rowdiv=rd)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    ops.gemm(a, b, k, n, rowdiv=rd)
                e1.record(); torch.cuda.synchronize()
                res.append((e0.elapsed_time(e1) / 30 * 1e3, tile, sp))
            except RuntimeError as e:
                res.append((9e9, tile, sp))
    os.environ.pop('RDM_GEMM_TUNE')
    for _ in range(3):
        ops.gemm(a, b, k, n, rowdiv=rd)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        ops.gemm(a, b, k, n, rowdiv=rd)
    e1.record(); torch.cuda.synchronize()
    auto = e0.elapsed_time(e1) / 30 * 1e3
    res.sort()
    print(f'M={m} K={k} N={n}: auto {auto:.1f} us | best ' + ', '.join(f'{us:.1f}us(t{t},s{sp})' for us, t, sp in res[:5]), flush=True)
