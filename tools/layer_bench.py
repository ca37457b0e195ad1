"""Times of one attention-layer application, GPU only (the launches of a variant are captured into a HIP graph and replayed):
the fused rdm_attention_layer (all phases / without projections / projections only) against the launches it replaces."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
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
    n0, n1 = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (431, 411)
    N = n0 + n1
    rng = np.random.default_rng(0)
    t = lambda *s, scale=1.0: torch.from_numpy((rng.normal(size=s) * scale).astype(np.float32)).cuda()
    q, x, kv, emb = t(N, 128), t(N, 128), t(N, 256), t(N, 64)
    qkv = t(N, 384)
    tail = (t(128, 128, scale=.09), t(128), t(128).abs() + .5, t(128), t(256, 128, scale=.09), t(256), t(128, 256, scale=.06), t(128),
            t(128).abs() + .5, t(128))
    wqkv, bqkv, wq, bq, wkv, bkv = t(384, 128, scale=.09), t(384), t(128, 128, scale=.09), t(128), t(256, 128, scale=.09), t(256)
    out, d384, d128, d256 = (torch.zeros((N, w), device='cuda') for w in (128, 384, 128, 256))
    hid = torch.zeros((N, 128), device='cuda')
    self_segs = [(0, n0, qkv[:n0, 128:256], qkv[:n0, 256:]), (n0, n1, qkv[n0:, 128:256], qkv[n0:, 256:])]
    cross_segs = [(0, n0, kv[n0:, :128], kv[n0:, 128:])]
    P = lambda w, b, dst, rope, bits: (w, b, dst, rope, bits)
    variants = {
        'fused self  A+B+C(q,kv)': lambda: ops.attention_layer(out=out, q=qkv, x=x, segments=self_segs, tail=tail, emb=emb,
                                                                projections=[P(wq, bq, d128, 0, 3), P(wkv, bkv, d256, 0, 2)]),
        'fused self  A+B': lambda: ops.attention_layer(out=out, q=qkv, x=x, segments=self_segs, tail=tail, emb=emb),
        'fused cross1 A+B+C(kv,qkv)': lambda: ops.attention_layer(out=out, q=q, x=x, segments=cross_segs, tail=tail, emb=emb,
                                                                   projections=[P(wkv, bkv, d256, 0, 1), P(wqkv, bqkv, d384, 256, 1)]),
        'fused cross1 A+B': lambda: ops.attention_layer(out=out, q=q, x=x, segments=cross_segs, tail=tail, emb=emb),
        'fused C only (qkv+rope, all rows)': lambda: ops.attention_layer(out=out, segments=[(0, N, None, None)], emb=emb,
                                                                          projections=[P(wqkv, bqkv, d384, 256, 1)], projections_only=True),
        'fused C only (q, all rows)': lambda: ops.attention_layer(out=out, segments=[(0, N, None, None)], emb=emb,
                                                                   projections=[P(wq, bq, d128, 0, 1)], projections_only=True),
        'old attention_self_pair': lambda: ops.attention_self_pair(qkv[:, :128], qkv[:, 128:256], qkv[:, 256:], n0, 4, out=hid),
        'old attention cross (ref rows)': lambda: ops.attention(q[:n0], kv[n0:, :128], kv[n0:, 128:], 4, out=hid[:n0]),
        'old tail (all rows)': lambda: ops.attention_tail(hid, x, *tail, out=out),
        'old tail (ref rows)': lambda: ops.attention_tail(hid[:n0], x[:n0], *tail, out=out[:n0]),
        'old gemm qkv': lambda: ops.gemm(x, wqkv.t().contiguous(), 128, 384, bias=bqkv, out=d384),
        'old rope': lambda: ops.rope(d384[:, :128], d384[:, 128:256], emb),
    }
    wt = wqkv.t().contiguous()
    variants['old gemm qkv'] = lambda: ops.gemm(x, wt, 128, 384, bias=bqkv, out=d384)
    for name, fn in variants.items():
        print(f'{name:40s} {timed(fn):7.2f} us', flush=True)
