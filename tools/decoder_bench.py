"""Graph-replayed timing of the decoder stages: rdm_decoder_stage (concatenation inside the GEMM's operand loads) against
upsample_concat + linear_group_norm / gemm, at the path's sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdmnet_amd import ops
from kpconv_bench import timed

if __name__ == '__main__':
    torch.manual_seed(0)
    for name, ns, m, c1, c2, n, norm in (('decoder3', 1310, 3879, 1024, 512, 512, True), ('decoder2', 3879, 10961, 512, 256, 257, False)):
        coarse, skip = torch.randn(ns, c1, device='cuda'), torch.randn(m, c2, device='cuda')
        idx = torch.randint(0, ns, (m, 70), device='cuda')
        k = c1 + c2
        w = torch.randn(k, (n + 3) // 4 * 4, device='cuda') / k ** 0.5
        bias, gamma, beta = torch.randn(n, device='cuda'), torch.rand(n, device='cuda') + 0.5, torch.randn(n, device='cuda')
        g = (gamma, beta, 32) if norm else ()
        t_new = timed(lambda: ops.decoder_stage(coarse, idx, skip, w, n, bias, *g, act=ops.ACT_LEAKY), reps=10)
        cat = ops.upsample_concat(coarse, idx, skip)
        t_cat = timed(lambda: ops.upsample_concat(coarse, idx, skip), reps=10)
        if norm:
            t_lin = timed(lambda: ops.linear_group_norm(cat, w, k, n, bias, gamma, beta, 32, act=ops.ACT_LEAKY), reps=10)
        else:
            t_lin = timed(lambda: ops.gemm(cat, w, k, n, bias=bias), reps=10)
        print(f'{name}: decoder_stage {t_new:.1f} us | upsample_concat {t_cat:.1f} + linear {t_lin:.1f} = {t_cat + t_lin:.1f} us', flush=True)
