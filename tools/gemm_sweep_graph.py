"""Tile/split sweep with GPU-only timing: the launches of one configuration are captured into a HIP graph (through
torch.cuda.graph) and replayed, which removes the ~15 us/call Python floor of gemm_sweep.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('RDM_LIB_PATH', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rdmnet_amd', 'librdmnet_hip_lab.so'))  # RDM_GEMM_TUNE lives in the lab build (make -C rdmnet_amd/csrc lab)
import torch
from rdmnet_amd import ops

N = [32000, 10961, 3879, 1310, 563]
shapes = [('kp1_2', N[0], 480, 32), ('kp2_1', N[1], 480, 32), ('kp2_2', N[1], 960, 64), ('kp3_1', N[2], 960, 64),
          ('kp3_2', N[2], 1920, 128), ('kp4_1', N[3], 1920, 128), ('kp4_2', N[3], 3840, 256), ('kp5_1', N[4], 3840, 256),
          ('kp5_2', N[4], 7680, 512),
          ('u0a', N[0], 64, 32), ('u0b', N[0], 32, 128), ('u0c', N[0], 64, 128), ('u1a', N[0], 128, 32), ('u1b', N[1], 32, 128),
          ('u1c', N[1], 128, 64), ('u1d', N[1], 64, 256), ('u1e', N[1], 128, 256), ('u1f', N[1], 256, 64), ('u2a', N[1], 256, 64),
          ('u2', N[2], 256, 128), ('u2b', N[2], 128, 512), ('u2c', N[2], 256, 512), ('u2d', N[2], 512, 128), ('u3a', N[2], 512, 128),
          ('u3', N[3], 512, 256), ('u3b', N[3], 256, 1024), ('u3c', N[3], 512, 1024), ('u3d', N[3], 1024, 256), ('u4a', N[3], 1024, 256),
          ('u4', N[4], 1024, 512), ('u4b', N[4], 512, 2048), ('u4c', N[4], 1024, 2048), ('u4d', N[4], 2048, 512),
          ('dec4', N[3], 1284, 1024), ('dec3', N[2], 1536, 512), ('dec2', N[1], 768, 257)]

def timed(a, b, k, n, rd, bias=None):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            ops.gemm(a, b, k, n, rowdiv=rd, bias=bias)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(20):
                ops.gemm(a, b, k, n, rowdiv=rd, bias=bias)
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(5):
            g.replay()
        e1.record(s); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 100 * 1e3

if __name__ == '__main__':
    tot_auto = tot_best = 0.0
    table = {}
    for name, m, k, n in shapes:
        a = torch.randn(m, k, device='cuda'); b = torch.randn(k, (n + 3) // 4 * 4, device='cuda'); rd = torch.ones(m, device='cuda')
        os.environ.pop('RDM_GEMM_TUNE', None)
        auto = timed(a, b, k, n, rd)
        res = []
        for tile in (1, 2, 3):
            for sp in (0, 1, 2, 3, 4, 6, 8, 12, 16):
                if sp > max(k // 64, 1) or (tile == 3 and n > 64):
                    continue
                os.environ['RDM_GEMM_TUNE'] = f'{tile},{sp}'
                try:
                    res.append((timed(a, b, k, n, rd), tile, sp))
                except RuntimeError:
                    pass
        res.sort()
        table[name] = {'m': m, 'k': k, 'n': n, 'auto': auto, 'configs': [[t, sp, us] for us, t, sp in res]}
        tot_auto += auto; tot_best += res[0][0]
        print(f'{name:6s} M={m:6d} K={k:5d} N={n:5d}: auto {auto:6.1f} us | ' + ', '.join(f'{us:.1f}(t{t},s{sp})' for us, t, sp in res[:4]), flush=True)
    print(f'sum auto {tot_auto:.0f} us, sum best {tot_best:.0f} us')
    if len(sys.argv) > 1:  # full table for refitting the dispatch model (tools/gemm_model_fit.py)
        import json
        json.dump(table, open(sys.argv[1], 'w'))
