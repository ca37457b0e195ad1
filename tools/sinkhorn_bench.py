"""Graph-replayed timing of rdm_sinkhorn at the path's size (256 patches of up to 128 x 128, 100 iterations)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdmnet_amd import ops
from tail_bench import timed

if __name__ == '__main__':
    torch.manual_seed(0)
    for fill in (0.5, 0.8, 1.0):
        b, m = 256, 128
        scores = torch.randn(b, m, m, device='cuda') * 3
        rm = (torch.rand(b, m, device='cuda') < fill).to(torch.uint8)
        cm = (torch.rand(b, m, device='cuda') < fill).to(torch.uint8)
        alpha = torch.ones(1, device='cuda')
        t = timed(lambda: ops.sinkhorn(scores, rm, cm, alpha, 100), reps=5)
        print(f'fill {fill}: {t:.1f} us')
