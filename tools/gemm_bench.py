"""Times rdm_gemm on the shapes of one 2x16k-point pair (run on the GPU box)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from rdmnet_amd import ops

N = [32000, 13795, 5289, 1900, 700]
shapes = [('kp1_1', N[0], 16, 64), ('kp1_2', N[0], 480, 32), ('kp2_1', N[1], 480, 32), ('kp2_2', N[1], 960, 64),
          ('kp3_1', N[2], 960, 64), ('kp3_2', N[2], 1920, 128), ('kp4_1', N[3], 1920, 128), ('kp4_2', N[3], 3840, 256),
          ('kp5_1', N[4], 3840, 256), ('kp5_2', N[4], 7680, 512),
          ('u0a', N[0], 64, 32), ('u0b', N[0], 32, 128), ('u0c', N[0], 64, 128), ('u1a', N[1], 128, 32), ('u1b', N[1], 32, 128),
          ('u1c', N[1], 128, 64), ('u1d', N[1], 64, 256), ('u1e', N[1], 128, 256), ('u2', N[2], 256, 128), ('u2b', N[2], 128, 512),
          ('u2c', N[2], 256, 512), ('u3', N[3], 512, 256), ('u3b', N[3], 256, 1024), ('u3c', N[3], 512, 1024),
          ('u4', N[4], 1024, 512), ('u4b', N[4], 512, 2048), ('u4c', N[4], 1024, 2048), ('u4d', N[4], 2048, 512),
          ('dec4', N[3], 1284, 1024), ('dec3', N[2], 1536, 512), ('dec2', N[1], 768, 257),
          ('t_in', 350, 2048, 128), ('t_qkv', 350, 128, 384), ('t_lin', 350, 128, 128), ('t_exp', 350, 128, 256),
          ('t_sq', 350, 256, 128), ('t_out', 350, 128, 256), ('vote1', 700, 256, 512), ('vote2', 700, 512, 256)]
tot = 0.0
for name, m, k, n in shapes:
    a = torch.randn(m, k, device='cuda')
    b = torch.randn(k, (n + 3) // 4 * 4, device='cuda')
    for _ in range(3):
        ops.gemm(a, b, k, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        ops.gemm(a, b, k, n)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    tot += us
    print(f'{name:7s} M={m:6d} K={k:5d} N={n:5d}  {us:8.1f} us  {2.0 * m * k * n / us / 1e6:7.2f} TFLOP/s')
print('sum us', tot)
