"""Lab: bits and time of the dense products under the library selected by RDM_LIB_PATH / the lab knobs in the environment.
Run it twice -- e.g. without knobs (every product on the 64 x 64 tile) and with RDM_GEMM_WIDE=1 (lab build) -- and diff
the sha1 columns: the wide form has to return the same bits.   python tools/gemm_wide_check.py [out.json]"""
import ctypes, hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from rdmnet_amd import _lib, ops

torch.manual_seed(0)
dev = 'cuda'
# (name, M, K, N): the path's products of one pair (level sizes 32000 / 10961 / 3879 / 1310 / 563) + edge shapes
shapes = [('KPConv 256->256 L3', 1310, 3840, 256), ('unary 512->2048 L4', 563, 512, 2048), ('KPConv 512->512 L4', 563, 7680, 512),
          ('KPConv 128->128 L2', 3879, 1920, 128), ('unary 256->1024 L3', 1310, 256, 1024), ('unary 2048->512 L4', 563, 2048, 512),
          ('unary 128->512 L2', 3879, 128, 512), ('unary 64->256 L1', 10961, 64, 256), ('shortcut 1024->2048 L4', 563, 1024, 2048),
          ('unary 1024->256 L3', 1310, 1024, 256), ('unary 512->128 L2', 3879, 512, 128), ('shortcut 64->128 L0', 32000, 64, 128),
          ('unary 32->128 L0', 32000, 32, 128), ('unary 256->128 L2', 3879, 256, 128), ('shortcut 256->512 L2', 3879, 256, 512),
          ('shortcut 128->256 L1', 10961, 128, 256), ('shortcut 512->1024 L3', 1310, 512, 1024), ('KPConv 256->256 L3->L4', 563, 3840, 256),
          ('KPConv 128->128 L2->L3', 1310, 1920, 128), ('in_proj 2048->128 L4', 563, 2048, 128), ('unary 32->128 L1', 10961, 32, 128),
          ('edge 300x36x132', 300, 36, 132), ('edge 257x1284x260', 257, 1284, 260), ('edge 129x68x1028', 129, 68, 1028)]


def sha(*ts):
    h = hashlib.sha1()
    for t in ts:
        h.update(t.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()[:12]


def timed(fn, reps=20):
    for _ in range(3):
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


def plan():
    p = (ctypes.c_int * 4)()
    _lib.lib().rdm_gemm_last_plan(p)
    return list(p)


rows = []
tot = {1: 0.0, 4: 0.0}
for name, m, k, n in shapes:
    npad, kpad = ops.pad4(n), ops.pad4(k)
    b = torch.zeros(kpad, npad, device=dev)
    b[:k, :n] = torch.randn(k, n, device=dev) / k ** 0.5
    bias = torch.randn(n, device=dev)
    gamma, beta = torch.randn(n, device=dev), torch.randn(n, device=dev)
    rec = {'name': name, 'm': m, 'k': k, 'n': n}
    for mult in (1, 4):
        a = torch.randn(m * mult, kpad, device=dev)
        a[:, k:] = 0
        rowdiv = torch.randint(1, 40, (m * mult,), device=dev).float()
        groups = 32 if n % 32 == 0 else 4
        out = ops.gemm(a, b, kpad, n, bias=bias, act=ops.ACT_LEAKY)
        pl = plan()
        y = ops.linear_group_norm(a, b, kpad, n, bias, gamma, beta, groups, rowdiv=rowdiv, act=ops.ACT_LEAKY)
        rec[f'sha_x{mult}'] = sha(out, y)
        rec[f'plan_x{mult}'] = pl
        t = timed(lambda: ops.gemm(a, b, kpad, n, bias=bias, act=ops.ACT_LEAKY, out=out))
        rec[f'us_x{mult}'] = t
        rec[f'tf_x{mult}'] = 2.0 * m * mult * k * n / t / 1e6
        if not name.startswith('edge'):
            tot[mult] += t
    rows.append(rec)
    print(f"{name:28s} M={m:6d} K={k:5d} N={n:5d}  x1 {rec['plan_x1']} {rec['us_x1']:7.1f} us {rec['tf_x1']:6.1f} TF {rec['sha_x1']}   "
          f"x4 {rec['plan_x4']} {rec['us_x4']:7.1f} us {rec['tf_x4']:6.1f} TF {rec['sha_x4']}", flush=True)

# the decoder's virtual [upsample | skip] operand (decoder3 / decoder2 shapes + an edge)
for name, m, mc, c1, c2, n, norm in (('decoder3 1536->512 L2', 3879, 1310, 1024, 512, 512, True), ('decoder2 768->257 L1', 10961, 3879, 512, 256, 257, False),
                                     ('decoder edge', 333, 100, 64, 36, 132, True)):
    rec = {'name': name, 'm': m, 'k': c1 + c2, 'n': n}
    for mult in (1, 4):
        coarse, skip = torch.randn(mc * mult, c1, device=dev), torch.randn(m * mult, c2, device=dev)
        idx = torch.randint(0, mc * mult + 3, (m * mult, 4), device=dev)  # (some rows point past the coarse level: zero rows)
        b = torch.zeros(c1 + c2, ops.pad4(n), device=dev)
        b[:, :n] = torch.randn(c1 + c2, n, device=dev) / (c1 + c2) ** 0.5
        bias, gamma, beta = torch.randn(n, device=dev), torch.randn(n, device=dev), torch.randn(n, device=dev)
        groups = 32 if n % 32 == 0 else 4
        fn = lambda: ops.decoder_stage(coarse, idx, skip, b, n, bias, gamma if norm else None, beta if norm else None, groups, act=ops.ACT_LEAKY)
        y = fn()
        rec[f'sha_x{mult}'], rec[f'plan_x{mult}'] = sha(y), plan()
        t = timed(fn)
        rec[f'us_x{mult}'] = t
        rec[f'tf_x{mult}'] = 2.0 * m * mult * (c1 + c2) * n / t / 1e6
        if 'edge' not in name:
            tot[mult] += t
    rows.append(rec)
    print(f"{name:28s} M={m:6d} K={c1 + c2:5d} N={n:5d}  x1 {rec['plan_x1']} {rec['us_x1']:7.1f} us {rec['tf_x1']:6.1f} TF {rec['sha_x1']}   "
          f"x4 {rec['plan_x4']} {rec['us_x4']:7.1f} us {rec['tf_x4']:6.1f} TF {rec['sha_x4']}   (whole stage: GEMM + GroupNorm)", flush=True)
print(f'sum (edges excluded): x1 {tot[1]:.0f} us, x4 {tot[4]:.0f} us')
if len(sys.argv) > 1:
    json.dump(rows, open(sys.argv[1], 'w'), indent=1)
