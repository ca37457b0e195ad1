import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from rdmnet_amd import ops
g = torch.Generator().manual_seed(0)
for c, m, ns, h in [(64, 16, 40, 20), (64, 50, 60, 30), (32, 16, 40, 20), (32, 700, 900, 65), (64, 5000, 6000, 63)]:
    s_pts = torch.randn(ns, 3, generator=g) * 2
    q_pts = s_pts[torch.randint(0, ns, (m,), generator=g)] + 0.1 * torch.randn(m, 3, generator=g)
    feats = torch.randn(ns, c, generator=g)
    idx = torch.randint(0, ns, (m, h), generator=g)
    kp = torch.randn(15, 3, generator=g)
    W = torch.randn(15, c, c, generator=g) / np.sqrt(15 * c)
    bias = torch.randn(c, generator=g)
    fd = ops.feat_empty(ns, c, 'cuda'); fd.copy_(feats)
    packed = torch.from_numpy(ops.kpconv_pack_weights(W.numpy())).cuda()
    print('launch', c, m, flush=True)
    out = ops.kpconv_fused(q_pts.cuda(), s_pts.cuda(), fd, ops.row_positive(fd), idx.cuda(), kp.cuda(), 1.7, packed, bias.cuda(), c)
    torch.cuda.synchronize()
    sp = s_pts.double(); rel = sp[idx] - q_pts.double()[:, None]
    infl = torch.clamp(1 - ((rel[:, :, None] - kp.double()) ** 2).sum(-1).sqrt() / 1.7, min=0)
    wf = torch.einsum('mhk,mhc->mkc', infl, feats.double()[idx]).reshape(m, 15 * c)
    nn = (feats.sum(1) > 0)[idx].sum(1).clamp(min=1).double()
    ref = wf @ W.double().reshape(15 * c, c) / nn[:, None] + bias.double()
    print('ok', c, m, float((out.cpu().double() - ref).abs().max()), flush=True)
