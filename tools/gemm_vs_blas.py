"""Lab: the path's large fp32 products on this library's tiled MFMA kernels (rdm_gemm) against the vendor library behind torch.mm
(hipBLASLt / rocBLAS, fp32), alone on the GPU, HIP-graph-free event timing of 30 back-to-back launches; rows x 1 (one pair) and
x 4 (what a lock-step group launches together; rdm_gemm then sees ONE product of 4 M rows, which is not how the group runs --
the group keeps each pair's plan -- so the x 4 column of rdm_gemm is only indicative).   python tools/gemm_vs_blas.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdmnet_amd import ops

shapes = [('KPConv 256->256 L3', 1310, 3840, 256), ('unary 512->2048 L4', 563, 512, 2048), ('KPConv 512->512 L4', 563, 7680, 512),
          ('KPConv 128->128 L2', 3879, 1920, 128), ('decoder3 1536->512 L2', 3879, 1536, 512), ('unary 256->1024 L3', 1310, 256, 1024),
          ('decoder4 1284->1024 L3', 1310, 1284, 1024), ('unary 2048->512 L4', 563, 2048, 512), ('unary 128->512 L2', 3879, 128, 512),
          ('unary 64->256 L1', 10961, 64, 256), ('decoder2 768->260 L1', 10961, 768, 260), ('shortcut 1024->2048 L4', 563, 1024, 2048)]


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


print('| product | M | K | N | rdm_gemm us | TF | torch.mm us | TF | rdm_gemm 4M us | TF | torch.mm 4M us | TF |')
print('|---|---|---|---|---|---|---|---|---|---|---|---|')
tot = [0.0] * 4
for name, m, k, n in shapes:
    row = []
    for mult in (1, 4):
        a = torch.randn(m * mult, k, device='cuda')
        b = torch.randn(k, n, device='cuda')
        out = torch.empty(m * mult, n, device='cuda')
        t_r = timed(lambda: ops.gemm(a, b, k, n, out=out)) if n % 4 == 0 else float('nan')
        t_t = timed(lambda: torch.mm(a, b, out=out))
        fl = 2.0 * m * mult * k * n
        row += [t_r, fl / t_r / 1e6, t_t, fl / t_t / 1e6]
    for i, j in enumerate((0, 2, 4, 6)):
        tot[i] += row[j]
    print(f'| {name} | {m} | {k} | {n} | ' + ' | '.join(f'{x:.1f}' for x in row) + ' |')
print(f'sums: rdm_gemm {tot[0]:.0f} us, torch.mm {tot[1]:.0f} us; 4 M rows: rdm_gemm {tot[2]:.0f} us, torch.mm {tot[3]:.0f} us')
