"""python tools/gemm_one.py M K N [reps] [rowdiv] -- repeated rdm_gemm launches of one shape (for rocprofv3 --pmc /
--kernel-trace).  `rowdiv` routes through the tiled kernels like the KPConv contraction does."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from rdmnet_amd import ops
m, k, n = (int(x) for x in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
a = torch.randn(m, k, device='cuda')
b = torch.randn(k, (n + 3) // 4 * 4, device='cuda')
rd = torch.ones(m, device='cuda') if len(sys.argv) > 5 else None
for _ in range(reps):
    ops.gemm(a, b, k, n, rowdiv=rd)
torch.cuda.synchronize()
