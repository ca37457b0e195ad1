"""Experimental tiles (RDM_GEMM_TUNE tile 4 = 128x64, 5 = 64x128, 6 = 128x128 k32, 7 = 256x64) against the shipped ones on
the deep products of the path; graph-replayed (tools/gemm_sweep_graph.py)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from gemm_sweep_graph import shapes, timed
deep = [s for s in shapes if s[0].startswith(('kp', 'dec')) or s[0] in ('u3c', 'u4b', 'u4c', 'u4d', 'u2c', 'u1e', 'u0c')]
tot = {}
for name, m, k, n in deep:
    a = torch.randn(m, k, device='cuda'); b = torch.randn(k, (n + 3) // 4 * 4, device='cuda'); rd = torch.ones(m, device='cuda')
    os.environ.pop('RDM_GEMM_TUNE', None)
    auto = timed(a, b, k, n, rd)
    best = {}
    for tile in (2, 4, 5, 6, 7):
        if (tile in (5, 6) and n < 128) or (tile in (4, 6, 7) and m < 128):
            continue
        for sp in (0, 2, 3, 4, 6, 8, 12, 16):
            if sp > max(k // 128, 1):
                continue
            os.environ['RDM_GEMM_TUNE'] = f'{tile},{sp}'
            try:
                us = timed(a, b, k, n, rd)
            except RuntimeError:
                continue
            if tile not in best or us < best[tile][0]:
                best[tile] = (us, sp)
    print(f'{name:6s} M={m:6d} K={k:5d} N={n:5d}: auto {auto:6.1f} | ' + ' '.join(f't{t}:{us:.1f}(s{sp})' for t, (us, sp) in sorted(best.items())), flush=True)
    tot['auto'] = tot.get('auto', 0) + auto
    tot['best'] = tot.get('best', 0) + min(v[0] for v in best.values())
    for t, (us, sp) in best.items():
        tot[t] = tot.get(t, 0) + us
print(tot)
