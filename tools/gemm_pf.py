"""Dispatch-chosen configuration of every product shape of the path under the current environment (RDM_GEMM_TUNE etc.);
graph-replayed timings as in gemm_sweep_graph.py.  Run once per setting and compare the sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from gemm_sweep_graph import shapes, timed

if __name__ == '__main__':
    tot = 0.0
    out = []
    for name, m, k, n in shapes:
        a = torch.randn(m, k, device='cuda'); b = torch.randn(k, (n + 3) // 4 * 4, device='cuda'); rd = torch.ones(m, device='cuda')
        us = timed(a, b, k, n, rd)
        tot += us
        out.append(f'{name}:{us:.1f}')
    print(' '.join(out))
    print(f'sum {tot:.0f} us')
