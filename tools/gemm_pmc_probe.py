"""Lab: one large product (4 pairs' rows of decoder3: 15516 x 1536 x 512) 20 times on rdm_gemm and on torch.mm, for a rocprofv3 --pmc pass."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdmnet_amd import ops
m, k, n = 15516, 1536, 512
a = torch.randn(m, k, device='cuda'); b = torch.randn(k, n, device='cuda'); out = torch.empty(m, n, device='cuda')
for _ in range(20):
    ops.gemm(a, b, k, n, out=out)
for _ in range(20):
    torch.mm(a, b, out=out)
torch.cuda.synchronize()
