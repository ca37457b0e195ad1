"""Transformer-sized products: the 32x32 K-split kernel (RDM_GEMM_TUNE unset or '0,0') against the tiled kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from rdmnet_amd import ops
from gemm_sweep_graph import timed  # noqa

shapes = [(563, 2048, 128), (819, 2048, 128), (563, 128, 384), (819, 128, 384), (563, 128, 256), (563, 256, 128),
          (282, 128, 256), (563, 128, 128), (700, 256, 512), (700, 512, 256), (1310, 256, 128)]
for m, k, n in shapes:
    a = torch.randn(m, k, device='cuda'); b = torch.randn(k, n, device='cuda'); bias = torch.randn(n, device='cuda')
    os.environ.pop('RDM_GEMM_TUNE', None)
    small = timed(a, b, k, n, None, bias)
    res = []
    for cfg in ('1,0', '2,0', '2,1', '2,2', '2,4', '2,8', '3,0'):
        if cfg.startswith('3') and n > 64:
            continue
        os.environ['RDM_GEMM_TUNE'] = cfg
        res.append((timed(a, b, k, n, None, bias), cfg))
    res.sort()
    print(f'M={m} K={k} N={n}: small {small:.1f} us | ' + ', '.join(f'{us:.1f}({c})' for us, c in res[:4]), flush=True)
