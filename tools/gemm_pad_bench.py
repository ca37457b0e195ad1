"""Does the leading dimension of A matter (L2 channel aliasing)?  Times rdm_gemm with A row stride K vs K+pad."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from rdmnet_amd import ops
shapes = [('kp3_2', 5289, 1920, 128), ('kp4_2', 1900, 3840, 256), ('kp5_1', 700, 3840, 256), ('kp5_2', 700, 7680, 512),
          ('dec3', 5289, 1536, 512), ('dec4', 1900, 1284, 1024), ('u4c', 700, 1024, 2048), ('kp2_2', 13795, 960, 64)]
for name, m, k, n in shapes:
    for pad in (0, 16, 32, 64):
        a = torch.randn(m, k + pad, device='cuda')[:, :k]
        b = torch.randn(k, n, device='cuda')
        for _ in range(3):
            ops.gemm(a, b, k, n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.gemm(a, b, k, n)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50
        print(f'{name:7s} pad={pad:3d}  {us:8.1f} us  {2.0 * m * k * n / us / 1e6:7.2f} TFLOP/s')
