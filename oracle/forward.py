"""TEST INFRASTRUCTURE ONLY -- CPU restatement (torch, fp32) of the reference's inference forward.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product path (rdmnet_amd/) never does.

Parity status: PINNED against golden vectors captured from the reference itself
(tests/golden/gen_golden.py imports /root/reference, loads the same synthetic state dict and dumps
every stage; tests/test_oracle_forward.py replays them).  The restatement is functional: a flat
`W` dict (reference state-dict names -> torch tensors) instead of nn.Modules, and every stage is a
free function so tests can teacher-force it with the reference's exact stage inputs.

Each function cites the reference lines it follows (paths relative to the reference root).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import native as _native

NEG_INF = 1e12  # the reference's finite "infinity" (modules/sinkhorn/learnable_sinkhorn.py:7)


# ----------------------------------------------------------------------------- a3: collate
def pyramid(points, lengths, cfg, impl=None):
    """geotransformer/utils/data.py:13-77 precompute_data_stack_mode.  points f32[N,3] (ref then
    src), lengths i64[2]; numpy in, dict of torch tensors out.  `impl` = object with
    grid_subsampling/radius_neighbors on numpy (default: the oracle's native restatement)."""
    impl = impl or _native.restatement()
    n_stages, limits = cfg.backbone.num_stages, cfg.neighbor_limits
    voxel, radius = cfg.backbone.init_voxel_size, cfg.backbone.init_radius
    P, L = [np.ascontiguousarray(points, np.float32)], [np.ascontiguousarray(lengths, np.int64)]
    for _ in range(1, n_stages):
        voxel *= 2  # data.py:23-28: doubled before the first use
        p, l = impl.grid_subsampling(P[-1], L[-1], np.float32(voxel))
        P.append(p), L.append(l)
    nb, sub, up = [], [], []
    for i in range(n_stages):
        nb.append(impl.radius_neighbors(P[i], P[i], L[i], L[i], np.float32(radius))[:, :limits[i]])
        if i < n_stages - 1:
            sub.append(impl.radius_neighbors(P[i + 1], P[i], L[i + 1], L[i], np.float32(radius))[:, :limits[i]])
            up.append(impl.radius_neighbors(P[i], P[i + 1], L[i], L[i + 1], np.float32(radius * 2))[:, :limits[i + 1]])
        radius *= 2
    t = torch.from_numpy
    return {
        'points': [t(p) for p in P], 'lengths': [t(l) for l in L],
        'neighbors': [t(np.ascontiguousarray(x)) for x in nb],
        'subsampling': [t(np.ascontiguousarray(x)) for x in sub],
        'upsampling': [t(np.ascontiguousarray(x)) for x in up],
        'features': torch.ones((P[0].shape[0], 1), dtype=torch.float32),  # kitti/dataset.py:187-188
        'batch_size': 1,
    }


# ----------------------------------------------------------------------------- a4/a5: KPConv blocks
def _gather_rows(x, idx, pad_value):
    """index_select with a padding row appended (modules/ops/index_select.py:4-31)."""
    x = torch.cat([x, torch.full_like(x[:1], pad_value)], 0)
    return x[idx.reshape(-1)].reshape(*idx.shape, x.shape[1])


def kpconv(W, name, s_feats, q_points, s_points, idx, sigma, chunk=4096):
    """modules/kpconv/kpconv.py:79-122.  Chunked over queries only to bound memory."""
    kp, weights, bias = W[name + '.kernel_points'], W[name + '.weights'], W[name + '.bias']
    out = []
    for a in range(0, idx.shape[0], chunk):
        ids = idx[a:a + chunk]
        rel = _gather_rows(s_points, ids, 1e6) - q_points[a:a + chunk, None, :]              # (m,H,3)  :91-93
        d2 = ((rel[:, :, None, :] - kp) ** 2).sum(3)                                          # (m,H,K)  :96-98
        infl = torch.clamp(1 - torch.sqrt(d2) / sigma, min=0.0).transpose(1, 2)               # (m,K,H)  :99-100
        nf = _gather_rows(s_feats, ids, 0.0)                                                  # (m,H,C)  :103-104
        wf = torch.matmul(infl, nf).permute(1, 0, 2)                                          # (K,m,C)  :105-108
        y = torch.matmul(wf, weights).sum(0)                                                  # (m,C')   :109-110
        n_valid = (nf.sum(-1) > 0.0).sum(-1).clamp(min=1)                                     # :113-115
        out.append(y / n_valid[:, None])
    return torch.cat(out, 0) + bias


def group_norm(W, name, x, groups):
    """modules/kpconv/modules.py:33-50: statistics span ALL rows (both clouds)."""
    y = F.group_norm(x.t().unsqueeze(0), groups, W[name + '.norm.weight'], W[name + '.norm.bias'], 1e-5)
    return y.squeeze(0).t()


def unary(W, name, x, groups, relu=True):
    """modules/kpconv/modules.py:53-83."""
    y = group_norm(W, name + '.norm', F.linear(x, W[name + '.mlp.weight'], W[name + '.mlp.bias']), groups)
    return F.leaky_relu(y, 0.1) if relu else y


def max_pool(x, idx):
    """modules/kpconv/functional.py:54-67 (the zero shadow row takes part in the max)."""
    return _gather_rows(x, idx, 0.0).max(1)[0]


def conv_block(W, name, x, q, s, idx, sigma, groups):
    """modules/kpconv/modules.py:104-146."""
    return F.leaky_relu(group_norm(W, name + '.norm', kpconv(W, name + '.KPConv', x, q, s, idx, sigma), groups), 0.1)


def residual_block(W, name, x, q, s, idx, sigma, groups, strided):
    """modules/kpconv/modules.py:149-225."""
    y = unary(W, name + '.unary1', x, groups) if (name + '.unary1.mlp.weight') in W else x
    y = kpconv(W, name + '.KPConv', y, q, s, idx, sigma)
    y = F.leaky_relu(group_norm(W, name + '.norm_conv', y, groups), 0.1)
    y = unary(W, name + '.unary2', y, groups, relu=False)
    sc = max_pool(x, idx) if strided else x
    if (name + '.unary_shortcut.mlp.weight') in W:
        sc = unary(W, name + '.unary_shortcut', sc, groups, relu=False)
    return F.leaky_relu(y + sc, 0.1)


def encoder(W, cfg, data, taps=None):
    """experiments/backbone.py:72-107."""
    from rdmnet_amd.weights import encoder_blocks, kpconv_sigma
    P, groups = data['points'], cfg.backbone.group_norm
    x, feats = data['features'], []
    for name, kind, _, _, lvl, strided in encoder_blocks(cfg):
        out_lvl = lvl + 1 if strided else lvl
        idx = data['subsampling'][lvl] if strided else data['neighbors'][lvl]
        if kind == 'conv':
            x = conv_block(W, 'encoder.' + name, x, P[out_lvl], P[lvl], idx, kpconv_sigma(cfg, lvl), groups)
        else:
            x = residual_block(W, 'encoder.' + name, x, P[out_lvl], P[lvl], idx, kpconv_sigma(cfg, lvl), groups, strided)
        if taps is not None:
            taps['encoder.' + name] = x
        if name.endswith('_3') or name == 'encoder1_2':
            feats.append(x)
    return feats


def decoder(W, cfg, feats, data):
    """experiments/backbone.py:118-151 (nearest upsample = column 0, functional.py:6-22)."""
    g, up = cfg.backbone.group_norm, data['upsampling']

    def nearest(x, idx):
        return torch.cat([x, torch.zeros_like(x[:1])], 0)[idx[:, 0]]

    l4 = unary(W, 'decoder.decoder4', torch.cat([nearest(feats[4], up[3]), feats[3]], 1), g)
    l3 = unary(W, 'decoder.decoder3', torch.cat([nearest(l4, up[2]), feats[2]], 1), g)
    return F.linear(torch.cat([nearest(l3, up[1]), feats[1]], 1), W['decoder.decoder2.mlp.weight'],
                    W['decoder.decoder2.mlp.bias'])


# ----------------------------------------------------------------------------- a7: 3DRoFormer
def rotary(x, emb):
    """rdmnet/thdroformer/thdroformer.py:56-85.  x (h,n,d), emb (h,n,d/2): theta = 2*pi*sigmoid(emb),
    each angle used for a pair (x0,x1) -> (x0 cos - x1 sin, x1 cos + x0 sin)."""
    rot = torch.stack([-x[..., 1::2], x[..., 0::2]], -1).reshape(x.shape)
    theta = torch.sigmoid(emb.repeat_interleave(2, dim=-1)) * 3.14159265359 * 2
    return x * torch.cos(theta) + rot * torch.sin(theta)


def _heads(x, h):
    return x.reshape(x.shape[0], h, -1).transpose(0, 1)  # (n, h*c) -> (h, n, c)


def dense_attention(q, k, v, bf16=False):
    """softmax(q k^T / sqrt(d)) v per head ([h, n, d] tensors; thdroformer.py:20-40 with k=None).  bf16=True is
    BASELINE.json configs[3]: operands of both contractions rounded to bf16, softmax and sums in fp32."""
    if bf16:
        q, k, v = (t.bfloat16().float() for t in (q, k, v))
    scores = torch.softmax(torch.einsum('hnd,hmd->hnm', q, k) / q.shape[-1] ** 0.5, dim=-1)
    if bf16:
        scores = scores.bfloat16().float()
    return torch.matmul(scores, v)


def attention_layer(W, p, x, mem, heads, emb=None, bf16=False):
    """RPEAttentionLayer (thdroformer.py:88-173) when emb is given, AttentionLayer
    (modules/transformer/vanilla_transformer.py:15-103) otherwise; then AttentionOutput
    (modules/transformer/output_layer.py:6-21)."""
    a = p + '.attention.attention'
    q = _heads(F.linear(x, W[a + '.proj_q.weight'], W[a + '.proj_q.bias']), heads)
    k = _heads(F.linear(mem, W[a + '.proj_k.weight'], W[a + '.proj_k.bias']), heads)
    v = _heads(F.linear(mem, W[a + '.proj_v.weight'], W[a + '.proj_v.bias']), heads)
    if emb is not None:
        e = _heads(emb, heads)
        q, k = rotary(q, e), rotary(k, e)
    hid = dense_attention(q, k, v, bf16).transpose(0, 1).reshape(x.shape[0], -1)
    hid = F.linear(hid, W[p + '.attention.linear.weight'], W[p + '.attention.linear.bias'])
    y = F.layer_norm(hid + x, (x.shape[1],), W[p + '.attention.norm.weight'], W[p + '.attention.norm.bias'])
    z = F.linear(F.relu(F.linear(y, W[p + '.output.expand.weight'], W[p + '.output.expand.bias'])),
                 W[p + '.output.squeeze.weight'], W[p + '.output.squeeze.bias'])
    return F.layer_norm(y + z, (x.shape[1],), W[p + '.output.norm.weight'], W[p + '.output.norm.bias'])


def thdroformer(W, name, ref_pts, src_pts, ref_x, src_x, num_layers, heads, bf16=False):
    """rdmnet/thdroformer/thdroformer.py:266-347; layer order and the sequential cross update
    follow RPEConditionalTransformer.forward :227-251."""
    e0 = F.linear(ref_pts, W[name + '.embedding.proj.weight'], W[name + '.embedding.proj.bias'])
    e1 = F.linear(src_pts, W[name + '.embedding.proj.weight'], W[name + '.embedding.proj.bias'])
    f0 = F.linear(ref_x, W[name + '.in_proj.weight'], W[name + '.in_proj.bias'])
    f1 = F.linear(src_x, W[name + '.in_proj.weight'], W[name + '.in_proj.bias'])
    for i in range(2 * num_layers):
        p = f'{name}.transformer.layers.{i}'
        if i % 2 == 0:
            f0 = attention_layer(W, p, f0, f0, heads, e0, bf16=bf16)
            f1 = attention_layer(W, p, f1, f1, heads, e1, bf16=bf16)
        else:
            f0 = attention_layer(W, p, f0, f1, heads, bf16=bf16)
            f1 = attention_layer(W, p, f1, f0, heads, bf16=bf16)  # sees the UPDATED f0 (:244-245)
    return (F.linear(f0, W[name + '.out_proj.weight'], W[name + '.out_proj.bias']),
            F.linear(f1, W[name + '.out_proj.weight'], W[name + '.out_proj.bias']))


# ----------------------------------------------------------------------------- a9/a10: vote + NMS
def vote(W, cfg, xyz, feats):
    """rdmnet/vote/vote.py:83-117."""
    x = feats
    for i in range(len(cfg.Vote.MLPS)):
        x = F.linear(x, W[f'vote.mlp_modules.{3 * i}.weight'], W[f'vote.mlp_modules.{3 * i}.bias'])
        x = F.relu(F.layer_norm(x, (x.shape[1],), W[f'vote.mlp_modules.{3 * i + 1}.weight'],
                                W[f'vote.mlp_modules.{3 * i + 1}.bias']))
    off = F.linear(x, W['vote.ctr_reg.weight'], W['vote.ctr_reg.bias'])
    lim = torch.tensor(cfg.Vote.MAX_TRANSLATE_RANGE, dtype=torch.float32)
    shift = torch.minimum(torch.maximum(off[:, :3], -lim), lim)  # two torch.where clamps, vote.py:104-106
    new_feats = F.layer_norm(feats + off[:, 3:], (feats.shape[1],), W['vote.out_proj.0.weight'], W['vote.out_proj.0.bias'])
    return xyz + shift, new_feats


def nms(nodes, lengths, radius, limit, impl=None):
    """rdmnet/vote/vote.py:13-40: radius search among the shifted nodes, then a greedy sweep in
    index order -- node i survives iff none of its (first `limit`) neighbours survived before it."""
    impl = impl or _native.restatement()
    idx = impl.radius_neighbors(nodes.numpy(), nodes.numpy(), lengths.numpy(), lengths.numpy(), np.float32(radius))[:, :limit]
    keep = np.zeros(idx.shape[0] + 1, dtype=bool)
    for i in range(idx.shape[0]):
        if not keep[idx[i]].any():
            keep[i] = True
    return torch.from_numpy(keep[:-1]), torch.from_numpy(np.ascontiguousarray(idx))


# ----------------------------------------------------------------------------- a11/a12: grouping, coarse matching
def sq_dist(x, y, normalized=False):
    """modules/ops/pairwise_distance.py:4-31."""
    xy = torch.matmul(x, y.transpose(-1, -2))
    if normalized:
        d = 2.0 - 2.0 * xy
    else:
        d = (x ** 2).sum(-1).unsqueeze(-1) - 2 * xy + (y ** 2).sum(-1).unsqueeze(-2)
    return d.clamp(min=1e-12)


def point_to_node(points, nodes, k):
    """modules/ops/pointcloud_partition.py:60-107."""
    d = sq_dist(nodes, points)                                   # (M,N)
    owner = d.min(0)[1]                                          # (N,)
    node_mask = torch.zeros(nodes.shape[0], dtype=torch.bool)
    node_mask[owner] = True
    own = torch.zeros_like(d, dtype=torch.bool)
    own[owner, torch.arange(points.shape[0])] = True
    d = d.masked_fill(~own, 1e12)
    knn = d.topk(k=k, dim=1, largest=False)[1]                   # (M,k)
    knn_mask = owner[knn] == torch.arange(nodes.shape[0])[:, None]
    knn = knn.masked_fill(~knn_mask, points.shape[0])
    return owner, node_mask, knn, knn_mask


def coarse_matching(ref_f, src_f, ref_mask, src_mask, k, dual=True):
    """modules/geotransformer/superpoint_matching.py:14-61."""
    ri, si = torch.nonzero(ref_mask, as_tuple=True)[0], torch.nonzero(src_mask, as_tuple=True)[0]
    s = torch.exp(-sq_dist(ref_f[ri], src_f[si], normalized=True))
    if dual:
        s = (s / s.sum(1, keepdim=True)) * (s / s.sum(0, keepdim=True))
    val, flat = s.reshape(-1).topk(k=min(k, s.numel()), largest=True)
    return ri[flat // s.shape[1]], si[flat % s.shape[1]], val


# ----------------------------------------------------------------------------- a14: Sinkhorn
def sinkhorn(scores, row_mask, col_mask, alpha, iters):
    """modules/sinkhorn/learnable_sinkhorn.py:13-66."""
    b, m, n = scores.shape
    prm = torch.zeros(b, m + 1, dtype=torch.bool)
    prm[:, :m] = ~row_mask
    pcm = torch.zeros(b, n + 1, dtype=torch.bool)
    pcm[:, :n] = ~col_mask
    z = torch.cat([torch.cat([scores, alpha.expand(b, m, 1)], -1), alpha.expand(b, 1, n + 1)], 1)
    z = z.masked_fill(prm[:, :, None] | pcm[:, None, :], -NEG_INF)
    nr, nc = row_mask.float().sum(1), col_mask.float().sum(1)
    norm = -torch.log(nr + nc)
    log_mu = torch.empty(b, m + 1)
    log_mu[:, :m] = norm[:, None]
    log_mu[:, m] = torch.log(nc) + norm
    log_mu[prm] = -NEG_INF
    log_nu = torch.empty(b, n + 1)
    log_nu[:, :n] = norm[:, None]
    log_nu[:, n] = torch.log(nr) + norm
    log_nu[pcm] = -NEG_INF
    u, v = torch.zeros_like(log_mu), torch.zeros_like(log_nu)
    for _ in range(iters):
        u = log_mu - torch.logsumexp(z + v[:, None, :], dim=2)
        v = log_nu - torch.logsumexp(z + u[:, :, None], dim=1)
    return z + u[:, :, None] + v[:, None, :] - norm[:, None, None]


# ----------------------------------------------------------------------------- a15/a16: LGR
def procrustes(src, ref, w, eps=1e-5):
    """modules/registration/procrustes.py:6-73 (batched (B,N,3); returns (B,4,4))."""
    w = torch.where(w < 0.0, torch.zeros_like(w), w)
    w = (w / (w.sum(1, keepdim=True) + eps))[:, :, None]
    cs, cr = (src * w).sum(1, keepdim=True), (ref * w).sum(1, keepdim=True)
    H = (src - cs).permute(0, 2, 1) @ (w * (ref - cr))
    U, _, V = torch.svd(H)
    eye = torch.eye(3).repeat(src.shape[0], 1, 1)
    eye[:, 2, 2] = torch.sign(torch.det(V @ U.transpose(1, 2)))
    R = V @ eye @ U.transpose(1, 2)
    t = cr.permute(0, 2, 1) - R @ cs.permute(0, 2, 1)
    T = torch.eye(4).repeat(src.shape[0], 1, 1)
    T[:, :3, :3], T[:, :3, 3] = R, t.squeeze(2)
    return T


def _apply(T, pts):
    """modules/ops/transformation.py:34-52."""
    if T.ndim == 2:
        return pts @ T[:3, :3].t() + T[:3, 3]
    return pts @ T[:, :3, :3].transpose(1, 2) + T[:, None, :3, 3]


def correspondence_matrix(score, ref_mask, src_mask):
    """local_global_registration.py:49-91 with k=1, dustbin, non-mutual (experiments/config.py:152-161)."""
    b, m, n = score.shape
    rv, ri = score.topk(k=1, dim=2)
    rmat = torch.zeros_like(score).scatter_(2, ri, rv)
    ref_side = rmat > score[:, :, -1:].expand(-1, -1, n)
    cv, ci = score.topk(k=1, dim=1)
    cmat = torch.zeros_like(score).scatter_(1, ci, cv)
    src_side = cmat > score[:, -1:, :].expand(-1, m, -1)
    corr = (ref_side | src_side)[:, :-1, :-1]
    return corr & (ref_mask[:, :, None] & src_mask[:, None, :])


def lgr(ref_knn, src_knn, ref_mask, src_mask, log_scores, cfg, force_best=None):
    """modules/geotransformer/local_global_registration.py:145-243.  `force_best` (test probe, not in the
    reference): refine from that local hypothesis instead of the argmax of the inlier counts -- used to enumerate
    the poses the reference would return if a near-tie between hypotheses fell the other way."""
    fm = cfg.fine_matching
    score = torch.exp(log_scores)
    corr = correspondence_matrix(score, ref_mask, src_mask)
    score = score[:, :-1, :-1] * corr.float()
    bi, ri, si = torch.nonzero(corr, as_tuple=True)
    ref_c, src_c, sc = ref_knn[bi, ri], src_knn[bi, si], score[bi, ri, si]
    cuts = [0] + (torch.nonzero(bi[1:] != bi[:-1], as_tuple=True)[0] + 1).tolist() + [bi.shape[0]]
    chunks = [(x, y) for x, y in zip(cuts[:-1], cuts[1:]) if y - x >= fm.correspondence_threshold]
    info = {'chunks': chunks}
    if chunks:
        width = max(y - x for x, y in chunks)
        bs, br, bw = (torch.zeros(len(chunks), width, 3), torch.zeros(len(chunks), width, 3),
                      torch.zeros(len(chunks), width))
        for c, (x, y) in enumerate(chunks):
            bs[c, :y - x], br[c, :y - x], bw[c, :y - x] = src_c[x:y], ref_c[x:y], sc[x:y]
        Ts = procrustes(bs, br, bw)
        res = torch.linalg.norm(ref_c[None] - _apply(Ts, src_c[None]), dim=2)
        inl = res < fm.acceptance_radius
        best = inl.sum(1).argmax() if force_best is None else torch.tensor(int(force_best))
        cur = sc * inl[best].float()
        info.update(hypotheses=Ts, inlier_counts=inl.sum(1), best=int(best))
    else:
        T0 = procrustes(src_c[None], ref_c[None], sc[None])[0]
        cur = sc * (torch.linalg.norm(ref_c - _apply(T0, src_c), dim=1) < fm.acceptance_radius).float()
    T = procrustes(src_c[None], ref_c[None], cur[None])[0]
    for _ in range(fm.num_refinement_steps - 1):
        cur = sc * (torch.linalg.norm(ref_c - _apply(T, src_c), dim=1) < fm.acceptance_radius).float()
        T = procrustes(src_c[None], ref_c[None], cur[None])[0]
    return ref_c, src_c, sc, T, info


# ----------------------------------------------------------------------------- a17: the forward
@torch.no_grad()
def forward(W, cfg, data, taps=None, impl=None):
    """experiments/model_infer.py:109-354 (inference).  Returns the output dict; `taps`
    (optional dict) receives the stage intermediates used by the teacher-forced tests."""
    taps = taps if taps is not None else {}
    out = {}
    t = cfg.thdroformer
    L = data['lengths']
    n_c, n_f, n_0 = int(L[-1][0]), int(L[1][0]), int(L[0][0])
    pts_c, pts_f, pts = data['points'][-1], data['points'][1], data['points'][0]
    out.update(ori_ref_points_c=pts_c[:n_c], ori_src_points_c=pts_c[n_c:], ref_points_f=pts_f[:n_f],
               src_points_f=pts_f[n_f:], ref_points=pts[:n_0], src_points=pts[n_0:])

    feats = encoder(W, cfg, data, taps)
    taps['feats_c_enc'] = feats[-1]
    bf16 = bool(getattr(t, 'attention_bf16', False))
    rf, sf = thdroformer(W, 'transformer', pts_c[:n_c], pts_c[n_c:], feats[-1][:n_c], feats[-1][n_c:],
                         t.num_layers, t.num_heads, bf16)
    taps['t1_ref'], taps['t1_src'] = rf, sf
    wn, bn = W['proj_n2p_score.weight'], W['proj_n2p_score.bias']
    r_n2p_logit, s_n2p_logit = F.linear(rf, wn, bn), F.linear(sf, wn, bn)
    r_n2p = torch.sigmoid(r_n2p_logit.view(-1)).clamp(0, 1)
    s_n2p = torch.sigmoid(s_n2p_logit.view(-1)).clamp(0, 1)
    feats[-1] = torch.cat([torch.cat([rf, r_n2p_logit], 1), torch.cat([sf, s_n2p_logit], 1)], 0)
    dec = decoder(W, cfg, feats, data)
    taps['decoder'] = dec
    feats_f, p2p = dec[:, :-1], dec[:, -1]
    out.update(ref_p2p_scores_c=torch.sigmoid(p2p[:n_f]).clamp(0, 1), src_p2p_scores_c=torch.sigmoid(p2p[n_f:]).clamp(0, 1))

    if cfg.Vote.model_use_vote and cfg.Vote.inference_use_vote:
        shifted, vfeats = vote(W, cfg, pts_c, torch.cat([rf, sf], 0))
        taps['vote_xyz'], taps['vote_feats'] = shifted, vfeats
        out.update(shifted_ref_points_c=shifted[:n_c], shifted_src_points_c=shifted[n_c:])
        w2, b2 = W['proj_n2n_score.weight'], W['proj_n2n_score.bias']
        n2n = torch.sigmoid(F.linear(vfeats, w2, b2).view(-1)).clamp(0, 1)
        keep, nms_idx = nms(shifted, L[-1], cfg.Vote.NMS_radius, cfg.neighbor_limits[-1], impl)
        taps['nms_mask'], taps['nms_idx'] = keep, nms_idx
        rk, sk = keep[:n_c], keep[n_c:]
        ref_c, src_c = shifted[:n_c][rk], shifted[n_c:][sk]
        out.update(ref_n2p_scores_c=r_n2p[rk], src_n2p_scores_c=s_n2p[sk], ref_n2n_scores_c=n2n[:n_c][rk],
                   src_n2n_scores_c=n2n[n_c:][sk], ref_points_c=ref_c, src_points_c=src_c)
        rf2, sf2 = thdroformer(W, 'transformer2', ref_c, src_c, vfeats[:n_c][rk], vfeats[n_c:][sk], t.num_layers2, t.num_heads,
                               bf16)
        taps['t2_ref'], taps['t2_src'] = rf2, sf2
    else:
        # infer.py:119-120 switches the vote layer off for Mulran, but model_infer.py:179-246 then never
        # defines ref_points_c (the reference raises).  Defined here as SURVEY.md §7 hard part 7 does:
        # superpoints = the un-shifted coarse points, features = the first transformer's output.
        ref_c, src_c, rf2, sf2 = pts_c[:n_c], pts_c[n_c:], rf, sf
        out.update(ref_n2p_scores_c=r_n2p, src_n2p_scores_c=s_n2p, ref_points_c=ref_c, src_points_c=src_c)
    rfn, sfn = F.normalize(rf2, p=2, dim=1), F.normalize(sf2, p=2, dim=1)
    out.update(ref_feats_c=rfn, src_feats_c=sfn)

    k = cfg.model.num_points_in_patch
    _, r_nmask, r_knn, r_kmask = point_to_node(pts_f[:n_f], ref_c, k)
    _, s_nmask, s_knn, s_kmask = point_to_node(pts_f[n_f:], src_c, k)
    taps.update(ref_node_masks=r_nmask, src_node_masks=s_nmask, ref_knn=r_knn, src_knn=s_knn,
                ref_knn_masks=r_kmask, src_knn_masks=s_kmask)
    out.update(ref_feats_f=feats_f[:n_f], src_feats_f=feats_f[n_f:])
    r_sel, s_sel, node_scores = coarse_matching(rfn, sfn, r_nmask, s_nmask, cfg.coarse_matching.num_correspondences,
                                                cfg.coarse_matching.dual_normalization)
    taps['node_corr_scores'] = node_scores
    out.update(ref_node_corr_indices=r_sel, src_node_corr_indices=s_sel)

    def pad0(x):
        return torch.cat([x, torch.zeros_like(x[:1])], 0)

    r_idx, s_idx = r_knn[r_sel], s_knn[s_sel]
    r_pts, s_pts = pad0(pts_f[:n_f])[r_idx], pad0(pts_f[n_f:])[s_idx]
    r_pm, s_pm = r_kmask[r_sel], s_kmask[s_sel]
    out.update(ref_node_corr_knn_points=r_pts, src_node_corr_knn_points=s_pts, ref_node_corr_knn_masks=r_pm,
               src_node_corr_knn_masks=s_pm)
    scores = torch.einsum('bnd,bmd->bnm', pad0(feats_f[:n_f])[r_idx], pad0(feats_f[n_f:])[s_idx]) / feats_f.shape[1] ** 0.5
    taps['patch_scores'] = scores
    ms = sinkhorn(scores, r_pm, s_pm, W['optimal_transport.alpha'], cfg.model.num_sinkhorn_iterations)
    out['matching_scores'] = ms
    rc, sc_, cs, T, info = lgr(r_pts, s_pts, r_pm, s_pm, ms, cfg)
    taps['lgr'] = info
    out.update(ref_corr_points=rc, src_corr_points=sc_, corr_scores=cs, estimated_transform=T)
    return out


def procrustes_fp64(src, ref, w):
    """weighted_procrustes (modules/registration/procrustes.py:6-73) in float64 numpy, returning also the
    singular values of the covariance.  Test-side probe: when the inlier set is (nearly) collinear the
    covariance is rank-deficient, the reference's fp32 torch.svd result is rounding noise, and a parity
    bound against it is meaningless; the tests then bound the HIP pose against this fp64 solution."""
    src, ref, w = (np.asarray(x, np.float64) for x in (src, ref, w))
    w = w / (w.sum() + 1e-5)
    cs, cr = (w[:, None] * src).sum(0), (w[:, None] * ref).sum(0)
    H = ((src - cs) * w[:, None]).T @ (ref - cr)
    U, S, Vt = np.linalg.svd(H)
    V = Vt.T
    R = V @ np.diag([1.0, 1.0, np.sign(np.linalg.det(V @ U.T))]) @ U.T
    T = np.eye(4)
    T[:3, :3], T[:3, 3] = R, cr - R @ cs
    return T, S


def to_torch(state):
    return {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in state.items()}


def rre_rte(T_est, T_ref):
    """Relative rotation error (deg) and translation error (m) between two 4x4 transforms.
    Same quantities as geotransformer/modules/registration/metrics.py:47-111, but the angle is taken
    from ||R_rel - I||_F = 2*sqrt(2)*sin(theta/2), which -- unlike acos((tr-1)/2) -- stays accurate
    for the micro-degree differences parity tests look at (acos amplifies one fp32 ulp to 0.02 deg)."""
    T_est, T_ref = np.asarray(T_est, np.float64), np.asarray(T_ref, np.float64)
    R = T_ref[:3, :3].T @ T_est[:3, :3]
    s = min(1.0, np.linalg.norm(R - np.eye(3)) / (2.0 * math.sqrt(2.0)))
    return math.degrees(2.0 * math.asin(s)), float(np.linalg.norm(T_ref[:3, 3] - T_est[:3, 3]))
