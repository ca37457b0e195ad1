"""Graph-replayed timings of the transformer-width fused kernels (attention tail, Linear + LayerNorm, 32x32 K-split GEMM)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdmnet_amd import ops


def timed(fn, reps=20):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(reps):
                fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


if __name__ == '__main__':
    r = lambda *s: torch.randn(*s, device='cuda')
    for m in (350, 700, 1126):
        hid, x = r(m, 128), r(m, 128)
        wo, w1, w2 = r(128, 128) / 11, r(256, 128) / 11, r(128, 256) / 16
        bo, b1, b2, g1, be1, g2, be2 = r(128), r(256), r(128), r(128), r(128), r(128), r(128)
        w1b = w1.t().contiguous()
        t_tail = timed(lambda: ops.attention_tail(hid, x, wo, bo, g1, be1, w1, b1, w2, b2, g2, be2))
        t_ln = timed(lambda: ops.linear_layer_norm(hid, wo, 128, 128, bo, g1, be1, residual=x))
        z = r(m, 256)
        t_ln2 = timed(lambda: ops.linear_layer_norm(z, w2, 256, 128, b2, g2, be2, residual=x))
        t_g = timed(lambda: ops.gemm(hid, w1b, 128, 256, bias=b1, act=1))
        print(f'm={m}: tail {t_tail:.1f} us | linear_ln k128 {t_ln:.1f}, k256 {t_ln2:.1f}, expand gemm {t_g:.1f} (sum {t_ln + t_ln2 + t_g:.1f})')
