"""GPU parity of the individual HIP operators (through the C-ABI) against fp32/fp64 torch CPU
restatements of the same reference formulas.  Tolerances are stated per test."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from rdmnet_amd import ops
    return ops


def padded(t, device='cuda'):
    """[n, c] tensor -> device view with row stride padded to a multiple of 4 (pads zeroed)."""
    n, c = t.shape
    ld = (c + 3) // 4 * 4
    buf = torch.zeros((n, ld), dtype=torch.float32, device=device)
    buf[:, :c] = t.to(device)
    return buf[:, :c]


@pytest.mark.parametrize('m,k,n,trans_b', [(1, 4, 1, False), (77, 36, 33, False), (300, 128, 257, False),
                                            (2236, 1284, 1024, False), (842, 7680, 512, False),
                                            (39609, 480, 32, False), (5000, 64, 128, False),
                                            (332, 256, 316, True), (129, 2048, 128, False)])
def test_gemm_matches_fp64(ops, m, k, n, trans_b):
    g = torch.Generator().manual_seed(m * 7 + n)
    a = torch.randn(m, k, generator=g)
    b = torch.randn(n, k, generator=g) if trans_b else torch.randn(k, n, generator=g)
    bias = torch.randn(n, generator=g)
    rowdiv = torch.randint(1, 9, (m,), generator=g).float()
    ref = (a.double() @ (b.double().t() if trans_b else b.double())) / rowdiv.double()[:, None] + bias.double()
    ref = F.leaky_relu(ref, 0.1)
    bd = b.cuda() if trans_b else padded(b)
    out = ops.gemm(padded(a), bd, k, n, trans_b=trans_b, bias=bias.cuda(), rowdiv=rowdiv.cuda(), act=ops.ACT_LEAKY)
    err = (out.cpu().double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= 2e-6 * scale * max(1.0, np.sqrt(k) / 8), (err, scale)  # fp32 accumulation over k terms


def test_gemm_batched_patch_scores(ops):
    g = torch.Generator().manual_seed(0)
    a, b = torch.randn(5, 128, 256, generator=g), torch.randn(5, 128, 256, generator=g)
    out = ops.gemm_batched(a.cuda(), b.cuda(), 256)
    ref = torch.einsum('bnd,bmd->bnm', a.double(), b.double())
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-4


@pytest.mark.parametrize('B,k,n_r,n_s,d', [(256, 128, 8000, 7500, 256), (3, 128, 50, 40, 256), (17, 64, 300, 300, 256),
                                         (9, 128, 200, 150, 32), (5, 64, 90, 80, 44)])  # d < 48: the shallow-K dispatch
def test_patch_scores_gathers_inside_the_gemm(ops, B, k, n_r, n_s, d):
    """rdm_patch_scores (model_infer.py:291-311: padded index_select of the patch features + einsum / sqrt(d)) against the
    separate launches (gather_rows + batched GEMM: same tile, same summation order -> same bits) and torch fp64 (1e-5 of the
    range).  Patches hold their real points first and shadow indices behind (point_to_node_partition), some none at all."""
    g = torch.Generator().manual_seed(B + k)
    rf, sf = torch.randn(n_r, d, generator=g), torch.randn(n_s, d, generator=g)

    def indices(n):
        idx = torch.randint(0, n, (B, k), generator=g)
        valid = torch.randint(0, k + 1, (B,), generator=g)
        valid[0] = 0
        valid[-1] = k
        return torch.where(torch.arange(k)[None] < valid[:, None], idx, torch.full_like(idx, n))
    ri, si = indices(n_r), indices(n_s)
    div = torch.full((k,), float(np.sqrt(d)))
    got = ops.patch_scores(padded(rf), ri.cuda(), padded(sf), si.cuda(), rowdiv=div.cuda())
    a = ops.gather_rows(padded(rf), ri.view(-1).cuda(), out=torch.empty((B * k, d), device='cuda')).view(B, k, d)
    b = ops.gather_rows(padded(sf), si.view(-1).cuda(), out=torch.empty((B * k, d), device='cuda')).view(B, k, d)
    sep = ops.gemm_batched(a, b, d, rowdiv=div.cuda())
    assert torch.equal(got, sep)
    rp, sp = torch.cat([rf, torch.zeros(1, d)]).double(), torch.cat([sf, torch.zeros(1, d)]).double()
    want = torch.einsum('bnd,bmd->bnm', rp[ri], sp[si]) / np.sqrt(d)
    assert (got.cpu().double() - want).abs().max() <= 1e-5 * want.abs().max()
    assert torch.count_nonzero(got[0]) == 0


@pytest.mark.parametrize('c,h,m,ns', [(1, 65, 500, 700), (32, 65, 400, 900), (64, 63, 300, 500), (128, 69, 200, 300),
                                      (256, 70, 150, 200), (512, 81, 90, 100), (32, 3, 50, 60),
                                      (96, 40, 120, 200), (160, 33, 70, 90), (1024, 20, 40, 60)])  # widths outside 32 * 2^k <= 512: the generic instance
def test_kpconv_gather_matches_reference_formula(ops, c, h, m, ns):
    """kpconv.py:91-105,113-115 restated in fp64 on random clouds with pad slots."""
    g = torch.Generator().manual_seed(c + h)
    s_pts = torch.randn(ns, 3, generator=g) * 2
    q_pts = s_pts[torch.randint(0, ns, (m,), generator=g)] + 0.1 * torch.randn(m, 3, generator=g)
    feats = torch.randn(ns, c, generator=g) if c > 1 else torch.ones(ns, 1)
    idx = torch.randint(0, ns, (m, h), generator=g)
    n_valid = torch.randint(0, h + 1, (m,), generator=g)
    idx = torch.where(torch.arange(h)[None] < n_valid[:, None], idx, torch.full_like(idx, ns))
    kp = torch.randn(15, 3, generator=g)
    sigma = 1.7
    sp = torch.cat([s_pts, torch.full((1, 3), 1e6)]).double()
    sf = torch.cat([feats, torch.zeros(1, c)]).double()
    rel = sp[idx] - q_pts.double()[:, None]
    infl = torch.clamp(1 - ((rel[:, :, None] - kp.double()) ** 2).sum(-1).sqrt() / sigma, min=0)
    ref = torch.einsum('mhk,mhc->mkc', infl, sf[idx]).reshape(m, 15 * c)
    pos = (feats.sum(1) > 0)
    nn_ref = torch.cat([pos, torch.zeros(1, dtype=torch.bool)])[idx].sum(1).clamp(min=1).float()
    wf, nn = ops.kpconv_gather(q_pts.cuda(), s_pts.cuda(), padded(feats), ops.row_positive(padded(feats)),
                               idx.cuda(), kp.cuda(), sigma)
    assert torch.equal(nn[:m].cpu(), nn_ref)
    got = wf.cpu().double()[:, :15 * c]
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize('c,h', [(1, 40), (32, 65), (64, 100), (128, 70)])
def test_kpconv_gather_shadow_slots_anywhere(ops, c, h):
    """The gather stops at a query's LAST real neighbour (the searches pad at the end of a row), so rows whose shadow slots sit
    at the front or in the middle, rows of shadow slots only and rows without any give what the row with its real
    neighbours packed to the front gives, bit for bit (each query accumulates its neighbours in slot order)."""
    g = torch.Generator().manual_seed(c * 1000 + h)
    ns, m = 500, 257
    s_pts = torch.randn(ns, 3, generator=g) * 2
    q_pts = s_pts[torch.randint(0, ns, (m,), generator=g)] + 0.1 * torch.randn(m, 3, generator=g)
    feats = torch.randn(ns, c, generator=g) if c > 1 else torch.ones(ns, 1)
    idx = torch.randint(0, ns, (m, h), generator=g)
    keep = torch.rand(m, h, generator=g) < 0.6
    keep[0] = False          # no neighbour at all
    keep[1] = True           # a full row
    keep[2, :-1] = False     # only the last slot
    keep[2, -1] = True
    scattered = torch.where(keep, idx, torch.full_like(idx, ns))
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)   # real neighbours first, order kept
    packed = torch.gather(scattered, 1, order)
    kp = torch.randn(15, 3, generator=g)
    args = (q_pts.cuda(), s_pts.cuda(), padded(feats), ops.row_positive(padded(feats)))
    wf_a, nn_a = ops.kpconv_gather(*args, scattered.cuda(), kp.cuda(), 1.3)
    wf_b, nn_b = ops.kpconv_gather(*args, packed.cuda(), kp.cuda(), 1.3)
    assert torch.equal(nn_a, nn_b)
    kdim = wf_a.shape[1] if c > 1 else 15
    if c > 1:  # (the four neighbours of an MFMA step are summed inside the instruction: groups shift when slots move)
        assert (wf_a - wf_b).abs().max() <= 1e-5 * wf_b.abs().max()
    else:
        assert (wf_a[:, :kdim] - wf_b[:, :kdim]).abs().max() <= 1e-5 * wf_b.abs().max()
    assert torch.count_nonzero(wf_a[0, :15 * c]) == 0


@pytest.mark.parametrize('c,h', [(1, 200), (32, 129), (64, 300), (256, 131)])
def test_kpconv_rows_wider_than_the_lds_staging(ops, c, h):
    """Neighbour limits beyond the 128 slots a wavefront stages in LDS run in chunks (the reference takes whatever
    calibrate_neighbors_stack_mode returns, utils/data.py:195-220; round 2 refused them): gather and one-kernel KPConv
    against the fp64 formula, 2e-5 of the range."""
    g = torch.Generator().manual_seed(7 * c + h)
    ns, m = 700, 150
    s_pts = torch.randn(ns, 3, generator=g) * 2
    q_pts = s_pts[torch.randint(0, ns, (m,), generator=g)] + 0.1 * torch.randn(m, 3, generator=g)
    feats = torch.randn(ns, c, generator=g) if c > 1 else torch.ones(ns, 1)
    idx = torch.randint(0, ns, (m, h), generator=g)
    n_valid = torch.randint(0, h + 1, (m,), generator=g)
    n_valid[0], n_valid[1] = h, 129
    idx = torch.where(torch.arange(h)[None] < n_valid[:, None], idx, torch.full_like(idx, ns))
    kp = torch.randn(15, 3, generator=g)
    sigma = 1.7
    sp = torch.cat([s_pts, torch.full((1, 3), 1e6)]).double()
    sf = torch.cat([feats, torch.zeros(1, c)]).double()
    rel = sp[idx] - q_pts.double()[:, None]
    infl = torch.clamp(1 - ((rel[:, :, None] - kp.double()) ** 2).sum(-1).sqrt() / sigma, min=0)
    ref = torch.einsum('mhk,mhc->mkc', infl, sf[idx]).reshape(m, 15 * c)
    nn_ref = torch.cat([feats.sum(1) > 0, torch.zeros(1, dtype=torch.bool)])[idx].sum(1).clamp(min=1).float()
    fd = padded(feats)
    wf, nn = ops.kpconv_gather(q_pts.cuda(), s_pts.cuda(), fd, ops.row_positive(fd), idx.cuda(), kp.cuda(), sigma)
    assert torch.equal(nn[:m].cpu(), nn_ref)
    assert (wf.cpu().double()[:, :15 * c] - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    cout = 64 if c == 1 else c
    if ops.kpconv_fused_supported(c, cout):
        W = torch.randn(15, c, cout, generator=g) / np.sqrt(15 * c)
        bias = torch.randn(cout, generator=g)
        want = ref @ W.double().reshape(15 * c, cout) / nn_ref.double()[:, None] + bias.double()
        packed = torch.from_numpy(ops.kpconv_pack_weights(W.numpy())).cuda()
        out = ops.kpconv_fused(q_pts.cuda(), s_pts.cuda(), fd, ops.row_positive(fd), idx.cuda(), kp.cuda(), sigma, packed, bias.cuda(), cout)
        assert (out.cpu().double() - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())


def test_kpconv_gather_refuses_channel_counts_it_has_no_layout_for(ops):
    """Feature widths must be 1 or a multiple of 32 (16 lanes x 8-byte pieces of a row): anything else is a RuntimeError
    with the reason, never a silent wrong answer (the reference's backbone widths are init_dim * 2^k, backbone.py:27-70)."""
    g = torch.Generator().manual_seed(1)
    s_pts, feats = torch.randn(50, 3, generator=g), torch.randn(50, 48, generator=g)
    idx = torch.randint(0, 50, (20, 8), generator=g)
    with pytest.raises(RuntimeError, match='unsupported channel count 48'):
        ops.kpconv_gather(s_pts[:20].cuda(), s_pts.cuda(), padded(feats), ops.row_positive(padded(feats)), idx.cuda(),
                          torch.randn(15, 3, generator=g).cuda(), 1.0)


def test_kpconv_gather_width_cap(ops):
    g = torch.Generator().manual_seed(3)
    ns, m, h, c = 200, 64, 20, 32
    s_pts, feats = torch.randn(ns, 3, generator=g), torch.randn(ns, c, generator=g)
    idx = torch.randint(0, ns, (m, h), generator=g)
    kp = torch.randn(15, 3, generator=g)
    args = (s_pts[:m].cuda(), s_pts.cuda(), padded(feats), ops.row_positive(padded(feats)))
    wf_full, _ = ops.kpconv_gather(*args, idx[:, :7].contiguous().cuda(), kp.cuda(), 1.0)
    wf_cap, _ = ops.kpconv_gather(*args, idx.cuda(), kp.cuda(), 1.0, width=torch.tensor([7], dtype=torch.int32).cuda())
    assert torch.equal(wf_full, wf_cap)


@pytest.mark.parametrize('n,c', [(5000, 32), (3001, 128), (842, 2048), (7, 64)])
def test_group_norm_matches_torch(ops, n, c):
    g = torch.Generator().manual_seed(n)
    x, res = torch.randn(n, c, generator=g) * 3 + 1, torch.randn(n, c, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    ref = F.group_norm(x.double().t()[None], 32, gamma.double(), beta.double(), 1e-5)[0].t()
    ref2 = F.leaky_relu(ref + res.double(), 0.1)
    y, pos = ops.group_norm(padded(x), gamma.cuda(), beta.cuda(), 32, act=ops.ACT_LEAKY, residual=padded(res),
                            want_positive=True)
    assert (y.cpu().double() - ref2).abs().max().item() <= 2e-5
    clear = ref2.sum(1).abs() > 1e-3
    assert torch.equal(pos[:n].cpu().bool()[clear], (ref2.sum(1) > 0)[clear])


@pytest.mark.parametrize('n,c', [(563, 2048), (563, 512), (1310, 1024), (1311, 256), (3879, 512), (4096, 128), (257, 64), (1, 64), (4097, 128)])
def test_group_norm_one_launch_form_has_the_bits_of_the_three_launches(ops, n, c):
    """On the coarse levels (up to 4 096 rows, whole 64-column slabs, no row flags) finalize and apply are one launch whose workgroups
    recompute the scale / shift of their slab the way the finalize kernel does: same bits as the separate launches, with and without
    residual and activation; and it matches torch's GroupNorm."""
    g = torch.Generator().manual_seed(7 * n + c)
    x, res = torch.randn(n, c, generator=g) * 2 - 0.5, torch.randn(n, c, generator=g)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    xd, rd, gd, bd = padded(x), padded(res), gamma.cuda(), beta.cuda()
    for act, r in ((ops.ACT_LEAKY, rd), (ops.ACT_NONE, None), (ops.ACT_RELU, rd)):
        one = ops.group_norm(xd, gd, bd, 32, act=act, residual=r)
        three = ops.group_norm(xd, gd, bd, 32, act=act, residual=r, form=1)
        assert torch.equal(one, three)
    ref = F.leaky_relu(F.group_norm(x.double().t()[None], 32, gamma.double(), beta.double(), 1e-5)[0].t() + res.double(), 0.1)
    assert (ops.group_norm(xd, gd, bd, 32, act=ops.ACT_LEAKY, residual=rd).cpu().double() - ref).abs().max().item() <= 2e-5


def test_layer_norm_matches_torch(ops):
    g = torch.Generator().manual_seed(5)
    for n, c in [(431, 128), (842, 512), (3, 256)]:
        x, res = torch.randn(n, c, generator=g), torch.randn(n, c, generator=g)
        gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
        ref = F.relu(F.layer_norm((x + res).double(), (c,), gamma.double(), beta.double()))
        y = ops.layer_norm(padded(x), gamma.cuda(), beta.cuda(), residual=padded(res), act=ops.ACT_RELU)
        assert (y.cpu().double() - ref).abs().max().item() <= 1e-5


def test_pooling_ops_are_exact(ops):
    g = torch.Generator().manual_seed(9)
    ns, m, h, c = 300, 120, 17, 256
    x = torch.randn(ns, c, generator=g)
    idx = torch.randint(0, ns + 1, (m, h), generator=g)  # includes pad index ns
    xp = torch.cat([x, torch.zeros(1, c)])
    assert torch.equal(ops.gather_max(padded(x), idx.cuda()).cpu(), xp[idx].max(1)[0])
    coarse, skip = torch.randn(ns, 257, generator=g), torch.randn(m, 1024, generator=g)
    y = ops.upsample_concat(padded(coarse), idx.cuda(), padded(skip))
    cp = torch.cat([coarse, torch.zeros(1, 257)])
    assert torch.equal(y.cpu(), torch.cat([cp[idx[:, 0]], skip], 1))
    assert y.stride(0) == 1284 and torch.count_nonzero(torch.as_strided(y, (m, 3), (1284, 1), 1281)) == 0
    # widths that are multiples of 4 take the 16-byte kernel (decoder3 / decoder2 of the path)
    coarse4, skip4 = torch.randn(ns, 512, generator=g), torch.randn(m, 256, generator=g)
    y4 = ops.upsample_concat(padded(coarse4), idx.cuda(), padded(skip4))
    assert torch.equal(y4.cpu(), torch.cat([torch.cat([coarse4, torch.zeros(1, 512)])[idx[:, 0]], skip4], 1))


@pytest.mark.parametrize('ns,m,c1,c2,n,norm', [(300, 1000, 64, 36, 128, True), (563, 1310, 257, 1024, 1024, True),
                                              (50, 1000, 32, 12, 64, True), (40, 300, 32, 4, 64, False),  # c1 + c2 < 48
                                              (1310, 3879, 1024, 512, 512, True), (900, 2500, 512, 256, 257, False)])
def test_decoder_stage_matches_torch(ops, ns, m, c1, c2, n, norm):
    """Decoder stage (backbone.py:118-151: nearest_upsample + cat + UnaryBlock / Linear) against torch fp64, 2e-5 of the output
    range; c1 % 32 == 0 takes the GEMM whose A tiles read the two sources directly, other widths the materialised rows --
    both must agree with the separate launches (upsample_concat + linear_group_norm / gemm) to the same tolerance."""
    g = torch.Generator().manual_seed(m + c1)
    coarse, skip = torch.randn(ns, c1, generator=g), torch.randn(m, c2, generator=g)
    idx = torch.randint(0, ns + 1, (m, 5), generator=g)  # includes the shadow index ns
    k = c1 + c2
    w, bias = torch.randn(k, n, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    gamma, beta = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    a = torch.cat([torch.cat([coarse, torch.zeros(1, c1)])[idx[:, 0]], skip], 1).double()
    lin = a @ w.double() + bias.double()
    want = F.leaky_relu(F.group_norm(lin.t()[None], 32, gamma.double(), beta.double(), 1e-5)[0].t(), 0.1) if norm else lin
    kp = (k + 3) // 4 * 4
    wp = torch.zeros(kp, n)
    wp[:k] = w
    args = (padded(coarse), idx.cuda(), padded(skip), padded(wp), n, bias.cuda())
    got = ops.decoder_stage(*args, gamma.cuda(), beta.cuda(), 32, act=ops.ACT_LEAKY) if norm else ops.decoder_stage(*args)
    assert (got.cpu().double() - want).abs().max() <= 2e-5 * want.abs().max()
    cat = ops.upsample_concat(padded(coarse), idx.cuda(), padded(skip))
    sep = (ops.linear_group_norm(cat, padded(wp), kp, n, bias.cuda(), gamma.cuda(), beta.cuda(), 32, act=ops.ACT_LEAKY) if norm
           else ops.gemm(cat, padded(wp), kp, n, bias=bias.cuda()))
    assert (got - sep).abs().max() <= 2e-5 * want.abs().max()


@pytest.mark.parametrize('m,k,n', [(5000, 64, 32), (13795, 128, 256), (700, 1024, 512), (77, 36, 64)])
def test_linear_group_norm_fused_matches_torch(ops, m, k, n):
    g = torch.Generator().manual_seed(m + n)
    x, w, bias = torch.randn(m, k, generator=g), torch.randn(k, n, generator=g) / k ** 0.5, torch.randn(n, generator=g)
    gamma, beta, res = torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g), torch.randn(m, n, generator=g)
    rowdiv = torch.randint(1, 5, (m,), generator=g).float()
    lin = (x.double() @ w.double()) / rowdiv.double()[:, None] + bias.double()
    ref = F.leaky_relu(F.group_norm(lin.t()[None], 32, gamma.double(), beta.double(), 1e-5)[0].t() + res.double(), 0.1)
    y = ops.linear_group_norm(padded(x), padded(w), k, n, bias.cuda(), gamma.cuda(), beta.cuda(), 32, rowdiv=rowdiv.cuda(),
                              act=ops.ACT_LEAKY, residual=padded(res))
    assert (y.cpu().double() - ref).abs().max().item() <= 5e-5


@pytest.mark.parametrize('m,n,lo,k', [(332, 316, -1.0, 256), (70, 90, 0.999, 256), (5, 7, 0.0, 256), (40, 60, 0.9995, 64)])
def test_coarse_matching_topk_paths(ops, m, n, lo, k):
    """Global top-k of the matching scores: descending, ties by ascending flat index.  Case 2/4 put thousands of scores
    into one histogram bin (more than the 2048-candidate list) -> the single-workgroup fallback; both must agree
    with a lexicographic sort of the matrix the call leaves behind."""
    rng = np.random.default_rng(m)
    sim = rng.uniform(lo, 1.0, (m, n)).astype(np.float32)
    sim[rng.integers(0, m, 50), rng.integers(0, n, 50)] = sim[0, 0]  # exact duplicates -> ties
    rmask = (rng.uniform(size=m) > 0.1).astype(np.uint8)
    cmask = (rng.uniform(size=n) > 0.1).astype(np.uint8)
    s = padded(torch.from_numpy(sim))
    ri, si, sc, cnt = ops.coarse_matching(s, torch.from_numpy(rmask).cuda(), torch.from_numpy(cmask).cuda(), k, dual=False)
    v = s.cpu().numpy()
    flat = v.reshape(-1)
    elig = np.flatnonzero(flat >= 0)
    order = elig[np.lexsort((elig, -flat[elig].astype(np.float64)))][:k]
    c = int(cnt)
    assert c == min(k, elig.size)
    assert np.array_equal(ri.cpu().numpy()[:c], order // n) and np.array_equal(si.cpu().numpy()[:c], order % n)
    assert np.array_equal(sc.cpu().numpy()[:c], flat[order])
    assert (v[rmask == 0] == -1).all() and (v[:, cmask == 0] == -1).all()


@pytest.mark.parametrize('m,k', [(1, 128), (431, 128), (563, 256), (77, 16), (1000, 2048)])
def test_linear_layer_norm_fused_matches_torch(ops, m, k):
    """LayerNorm(x W + b + residual) in one launch (transformer width 128) against torch fp64; tolerance 2e-5 of the
    output range (fp32 accumulation over k, statistics over 128 columns)."""
    rng = np.random.default_rng(m + k)
    x = torch.from_numpy(rng.normal(size=(m, k)).astype(np.float32))
    w = torch.from_numpy((rng.normal(size=(k, 128)) / np.sqrt(k)).astype(np.float32))
    b = torch.from_numpy(rng.normal(size=128).astype(np.float32))
    res = torch.from_numpy(rng.normal(size=(m, 128)).astype(np.float32))
    gamma = torch.from_numpy(rng.uniform(0.5, 1.5, 128).astype(np.float32))
    beta = torch.from_numpy(rng.normal(size=128).astype(np.float32))
    for residual in (res, None):
        h = x.double() @ w.double() + b.double() + (residual.double() if residual is not None else 0)
        want = F.layer_norm(h, (128,), gamma.double(), beta.double(), 1e-5)
        got = ops.linear_layer_norm(padded(x), w.t().contiguous().cuda(), k, 128, b.cuda(), gamma.cuda(), beta.cuda(),
                                    residual=padded(residual) if residual is not None else None).cpu()
        assert (got.double() - want).abs().max() <= 2e-5 * want.abs().max()
    relu = ops.linear_layer_norm(padded(x), w.t().contiguous().cuda(), k, 128, b.cuda(), gamma.cuda(), beta.cuda(), act=1).cpu()
    assert relu.min() >= 0
    with pytest.raises(RuntimeError):
        ops.linear_layer_norm(padded(x), w.t().contiguous().cuda(), k, 64, b.cuda(), gamma.cuda(), beta.cuda())


@pytest.mark.parametrize('m', [1, 16, 431, 700, 1001])
def test_attention_tail_fused_matches_torch(ops, m):
    """Output projection + residual LayerNorm + FFN + residual LayerNorm in one launch (thdroformer.py:142-173,
    output_layer.py:6-21) against torch fp64; tolerance 3e-5 of the output range (three chained fp32 products and
    two LayerNorms over 128 columns)."""
    rng = np.random.default_rng(m)
    t = lambda *s, scale=1.0: torch.from_numpy((rng.normal(size=s) * scale).astype(np.float32))
    hid, x = t(m, 128), t(m, 128)
    wo, w1, w2 = t(128, 128, scale=128 ** -0.5), t(256, 128, scale=128 ** -0.5), t(128, 256, scale=256 ** -0.5)
    bo, b1, b2 = t(128), t(256), t(128)
    g1, g2 = (torch.from_numpy(rng.uniform(0.5, 1.5, 128).astype(np.float32)) for _ in range(2))
    be1, be2 = t(128), t(128)
    y = F.layer_norm(hid.double() @ wo.double().t() + bo.double() + x.double(), (128,), g1.double(), be1.double(), 1e-5)
    z = torch.relu(y @ w1.double().t() + b1.double())
    want = F.layer_norm(z @ w2.double().t() + b2.double() + y, (128,), g2.double(), be2.double(), 1e-5)
    c = lambda v: v.cuda()
    got = ops.attention_tail(padded(hid), padded(x), c(wo), c(bo), c(g1), c(be1), c(w1), c(b1), c(w2), c(b2), c(g2), c(be2)).cpu()
    assert got.shape == (m, 128)
    assert (got.double() - want).abs().max() <= 3e-5 * want.abs().max()
    # the same through the three separate launches it replaces
    y3 = ops.linear_layer_norm(padded(hid), c(wo), 128, 128, c(bo), c(g1), c(be1), residual=padded(x))
    z3 = ops.gemm(y3, c(w1.t().contiguous()), 128, 256, bias=c(b1), act=1)
    o3 = ops.linear_layer_norm(z3, c(w2), 256, 128, c(b2), c(g2), c(be2), residual=y3).cpu()
    assert (got - o3).abs().max() <= 3e-5 * want.abs().max()
    # round 6: the weights in the kernel's operand order (what the engine and the module keep): the same bits
    packed = ops.attention_tail_pack_weights(c(wo), c(w1), c(w2))
    got_p = ops.attention_tail_packed(padded(hid), padded(x), packed, c(bo), c(g1), c(be1), c(b1), c(b2), c(g2), c(be2)).cpu()
    assert torch.equal(got_p, got)


@pytest.mark.parametrize('c,cout,h,m,ns', [(1, 64, 65, 1000, 1500), (32, 32, 65, 1000, 2000), (64, 64, 63, 700, 900), (32, 32, 3, 50, 60),
                                           (64, 64, 70, 16, 40), (1, 64, 9, 1, 5), (32, 32, 65, 65, 300)])
def test_kpconv_fused_matches_reference_formula(ops, c, cout, h, m, ns):
    """The one-kernel KPConv (rdm_kpconv_fused: aggregation + kernel-weight contraction + count normalisation + bias,
    kpconv.py:79-122) against the fp64 formula, on random clouds with pad slots, row counts that do not fill the last
    workgroup, and a width cap; its GroupNorm partials against the column sums of its own output."""
    g = torch.Generator().manual_seed(100 * c + h + m)
    s_pts = torch.randn(ns, 3, generator=g) * 2
    q_pts = s_pts[torch.randint(0, ns, (m,), generator=g)] + 0.1 * torch.randn(m, 3, generator=g)
    feats = torch.randn(ns, c, generator=g) if c > 1 else torch.ones(ns, 1)
    idx = torch.randint(0, ns, (m, h), generator=g)
    n_valid = torch.randint(0, h + 1, (m,), generator=g)
    idx = torch.where(torch.arange(h)[None] < n_valid[:, None], idx, torch.full_like(idx, ns))
    kp = torch.randn(15, 3, generator=g)
    W = torch.randn(15, c, cout, generator=g) / np.sqrt(15 * c)
    bias = torch.randn(cout, generator=g)
    sigma = 1.7
    sp = torch.cat([s_pts, torch.full((1, 3), 1e6)]).double()
    sf = torch.cat([feats, torch.zeros(1, c)]).double()
    rel = sp[idx] - q_pts.double()[:, None]
    infl = torch.clamp(1 - ((rel[:, :, None] - kp.double()) ** 2).sum(-1).sqrt() / sigma, min=0)
    wf = torch.einsum('mhk,mhc->mkc', infl, sf[idx]).reshape(m, 15 * c)
    pos = (feats.sum(1) > 0)
    nn_ref = torch.cat([pos, torch.zeros(1, dtype=torch.bool)])[idx].sum(1).clamp(min=1).double()
    ref = wf @ W.double().reshape(15 * c, cout) / nn_ref[:, None] + bias.double()
    packed = torch.from_numpy(ops.kpconv_pack_weights(W.numpy())).cuda()
    fd = padded(feats)
    args = (q_pts.cuda(), s_pts.cuda(), fd, ops.row_positive(fd), idx.cuda(), kp.cuda(), sigma, packed, bias.cuda(), cout)
    out, part = ops.kpconv_fused(*args, want_partials=True)
    got = out.cpu().double()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # partials: [blocks, 2, cout] fp64 sums / sums of squares of the rows of each block
    assert (part[:, 0].sum(0).cpu() - got.sum(0)).abs().max().item() <= 1e-9 * max(1.0, got.abs().sum(0).max().item())
    assert (part[:, 1].sum(0).cpu() - (got * got).sum(0)).abs().max().item() <= 1e-9 * max(1.0, (got * got).sum(0).max().item())
    # the same rows through the two-kernel form (gather + GEMM): same numbers up to the summation order
    wf2, nn2 = ops.kpconv_gather(*args[:6], sigma)
    kdim = 16 if c == 1 else 15 * c
    b = torch.zeros((ops.pad4(kdim), ops.pad4(cout)))
    b[:15 * c, :cout] = W.reshape(15 * c, cout)
    two = ops.gemm(wf2, b.cuda(), ops.pad4(kdim), cout, bias=bias.cuda(), rowdiv=nn2)
    assert (two.cpu().double() - got).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item())
    # width cap = the reference's [:, :min(limit, max_count)]
    if h > 4:
        cap = torch.tensor([h - 3], dtype=torch.int32).cuda()
        a = ops.kpconv_fused(*args, width=cap)
        b2 = ops.kpconv_fused(*args[:4], idx[:, :h - 3].contiguous().cuda(), *args[5:])
        assert torch.equal(a, b2)


@pytest.mark.parametrize('c,h,m,ns,spread', [(32, 65, 1000, 4000, 0.3), (64, 63, 777, 3000, 0.3), (32, 128, 333, 900, 3.0),
                                             (64, 100, 300, 5000, 3.0), (32, 17, 16, 40, 1.0), (64, 9, 5, 9, 1.0)])
def test_kpconv_lds_tile_kernel_has_the_bits_of_the_lock_step_kernel(ops, c, h, m, ns, spread):
    """Round 4, 'LDS-staged neighbour tiles': form 2 of rdm_kpconv_fused_form stages the union of the support rows of 16
    queries once in LDS and aggregates from there.  Neighbour order, influence arithmetic, MFMA sequence and the K split of the
    contraction are the lock-step kernel's (form 1), so the convolution output is the same BITS -- with the queries in row
    order, in a spatial order (shuffled order records, as rdm_radius_grid_records would give), with unions far beyond the tile's
    capacity (neighbours drawn from the whole cloud: most rows fall back to global fetches), shadow slots, a width cap and a
    last block that is not full; the GroupNorm partials are the column sums of the output, one row per 16 queries."""
    g = torch.Generator().manual_seed(1000 * c + h + m)
    s_pts = torch.randn(ns, 3, generator=g) * 2
    q_sel = torch.randint(0, ns, (m,), generator=g)
    q_pts = s_pts[q_sel] + 0.1 * torch.randn(m, 3, generator=g)
    feats = torch.randn(ns, c, generator=g)
    feats[torch.rand(ns, generator=g) < 0.2] *= -1  # (some rows with a negative sum: the positive flags matter)
    # neighbours: the h nearest support points of a jittered copy of the query (small spread: neighbouring queries share them;
    # large spread: nearly disjoint unions), padded behind a random count
    d = torch.cdist(q_pts + spread * torch.randn(m, 3, generator=g), s_pts)
    idx = d.argsort(1)[:, :h].contiguous()
    n_valid = torch.randint(0, h + 1, (m,), generator=g)
    n_valid[0] = h
    idx = torch.where(torch.arange(h)[None] < n_valid[:, None], idx, torch.full_like(idx, ns))
    kp = torch.randn(15, 3, generator=g)
    W = torch.randn(15, c, c, generator=g) / np.sqrt(15 * c)
    bias = torch.randn(c, generator=g)
    packed = torch.from_numpy(ops.kpconv_pack_weights(W.numpy())).cuda()
    fd = padded(feats)
    args = (q_pts.cuda(), s_pts.cuda(), fd, ops.row_positive(fd), idx.cuda(), kp.cuda(), 1.7, packed, bias.cuda(), c)
    lock = ops.kpconv_fused(*args, form=1)
    tile, part = ops.kpconv_fused(*args, form=2, want_partials=True)
    assert torch.equal(tile, lock)
    assert part.shape[0] == (m + 15) // 16
    got = tile.cpu().double()
    assert (part[:, 0].sum(0).cpu() - got.sum(0)).abs().max().item() <= 1e-9 * max(1.0, got.abs().sum(0).max().item())
    assert (part[:, 1].sum(0).cpu() - (got * got).sum(0)).abs().max().item() <= 1e-9 * max(1.0, (got * got).sum(0).max().item())
    # cell-ordered records {x, y, z, row}: here sorted along x (any permutation must give the same output)
    perm = torch.argsort(q_pts[:, 0])
    rec = torch.cat([q_pts[perm], perm.to(torch.int32).view(torch.float32)[:, None]], 1).contiguous().cuda()
    ordered, part_o = ops.kpconv_fused(*args, form=2, want_partials=True, order=rec)
    assert torch.equal(ordered, lock)
    rows16 = got[perm[:16]] if m >= 16 else got[perm]
    assert (part_o[0, 0].cpu() - rows16.sum(0)).abs().max().item() <= 1e-9 * max(1.0, rows16.abs().sum(0).max().item())
    # the default form is the tile kernel for these shapes
    assert torch.equal(ops.kpconv_fused(*args, order=rec), lock)
    if h > 4:
        cap = torch.tensor([h - 3], dtype=torch.int32).cuda()
        assert torch.equal(ops.kpconv_fused(*args, width=cap, form=2, order=rec), ops.kpconv_fused(*args, width=cap, form=1))


@pytest.mark.parametrize('c,h,m,ns,spread', [(128, 69, 700, 3000, 0.3), (256, 70, 333, 2000, 0.3), (512, 81, 130, 900, 0.5),
                                             (128, 128, 100, 4000, 3.0), (192, 17, 16, 40, 1.0), (256, 9, 5, 9, 1.0)])
def test_kpconv_gather_lds_tile_form_has_the_bits_of_the_per_neighbour_form(ops, c, h, m, ns, spread):
    """Round 5: form 2 of rdm_kpconv_gather_form serves the layers whose weight contraction stays a GEMM (c_in >= 128) with the
    aggregation half of the KPConv tile kernel -- a workgroup stages the union of the support rows of 16 ordered queries once in
    LDS, per 64-channel slice -- and writes WF and the positive-neighbour counts where kpconv_gather_kernel (form 1) writes
    them: same neighbour order, influences and MFMA sequence per accumulator, i.e. the same BITS -- for every channel count that
    is a multiple of 64, shadow slots, unions beyond the tile's capacity, a width cap, a ragged last block, any query order."""
    g = torch.Generator().manual_seed(7 * c + h + m)
    s_pts = torch.randn(ns, 3, generator=g) * 2
    q_pts = s_pts[torch.randint(0, ns, (m,), generator=g)] + 0.1 * torch.randn(m, 3, generator=g)
    feats = torch.randn(ns, c, generator=g)
    feats[torch.rand(ns, generator=g) < 0.2] *= -1
    d = torch.cdist(q_pts + spread * torch.randn(m, 3, generator=g), s_pts)
    idx = d.argsort(1)[:, :h].contiguous()
    n_valid = torch.randint(0, h + 1, (m,), generator=g)
    n_valid[0] = h
    idx = torch.where(torch.arange(h)[None] < n_valid[:, None], idx, torch.full_like(idx, ns))
    kp = torch.randn(15, 3, generator=g)
    fd = padded(feats)
    args = (q_pts.cuda(), s_pts.cuda(), fd, ops.row_positive(fd), idx.cuda(), kp.cuda(), 1.7)
    wf1, nn1 = ops.kpconv_gather(*args, form=1)
    perm = torch.argsort(q_pts[:, 0])
    rec = torch.cat([q_pts[perm], perm.to(torch.int32).view(torch.float32)[:, None]], 1).contiguous().cuda()
    ident = torch.arange(m)
    rec_rows = torch.cat([q_pts, ident.to(torch.int32).view(torch.float32)[:, None]], 1).contiguous().cuda()
    for order in (rec, rec_rows):
        wf2, nn2 = ops.kpconv_gather(*args, order=order, form=2)
        assert torch.equal(wf2[:, :15 * c], wf1[:, :15 * c]) and torch.equal(nn2[:m], nn1[:m])
    if h > 4:
        cap = torch.tensor([h - 3], dtype=torch.int32).cuda()
        a, na = ops.kpconv_gather(*args, width=cap, order=rec, form=2)
        b, nb = ops.kpconv_gather(*args, width=cap, form=1)
        assert torch.equal(a[:, :15 * c], b[:, :15 * c]) and torch.equal(na[:m], nb[:m])


@pytest.mark.parametrize('m,k,n', [(563, 7680, 512), (1310, 3840, 256), (3879, 1920, 128), (10961, 64, 256), (3879, 128, 512), (300, 36, 132),
                                   (257, 1284, 260), (129, 68, 1028), (32000, 32, 128)])
def test_gemm_forms_return_the_same_bits(ops, m, k, n):
    """Round 6: the WIDE forms of the tiled GEMM (rdm_gemm_form 1 / 2: 128 x 128 x 32 tiles with fragment-ordered LDS images, on
    v_mfma_f32_32x32x2_f32 or v_mfma_f32_16x16x4_f32) return the bits of the library's choice (form 0: the 64 x 64 tile) -- an
    output element is one fp32 fma chain over ascending k inside the same split-K ranges whatever the tile and the MFMA shape,
    and the GroupNorm column partials keep their 64-row blocks and their combination order.  Products with and without split-K,
    with bias / activation / row divisor / statistics, ragged edges in M, N and K."""
    g = torch.Generator().manual_seed(m + k + n)
    kp, n_p = (k + 3) // 4 * 4, (n + 3) // 4 * 4
    a = torch.zeros(m, kp)
    a[:, :k] = torch.randn(m, k, generator=g)
    w = torch.zeros(kp, n_p)
    w[:k, :n] = torch.randn(k, n, generator=g) / k ** 0.5
    bias, gamma, beta = torch.randn(n, generator=g), torch.rand(n, generator=g) + 0.5, torch.randn(n, generator=g)
    rowdiv = torch.randint(1, 40, (m,), generator=g).float()
    a, w, bias, gamma, beta, rowdiv = (t.cuda() for t in (a, w, bias, gamma, beta, rowdiv))
    groups = 32 if n % 32 == 0 else 4
    want_plain = ops.gemm(a, w, kp, n, bias=bias, act=ops.ACT_LEAKY)
    want_gn = ops.linear_group_norm(a, w, kp, n, bias, gamma, beta, groups, rowdiv=rowdiv, act=ops.ACT_LEAKY)
    for form in (1, 2):
        assert torch.equal(ops.gemm(a, w, kp, n, bias=bias, act=ops.ACT_LEAKY, form=form), want_plain), form
        assert torch.equal(ops.linear_group_norm(a, w, kp, n, bias, gamma, beta, groups, rowdiv=rowdiv, act=ops.ACT_LEAKY, form=form), want_gn), form


@pytest.mark.parametrize('ns,m,c1,c2,n,norm', [(1310, 3879, 1024, 512, 512, True), (900, 2500, 512, 256, 257, False), (100, 333, 64, 36, 132, True)])
def test_decoder_stage_forms_return_the_same_bits(ops, ns, m, c1, c2, n, norm):
    """The decoder's virtual [upsample | skip] operand on the wide forms (rdm_decoder_stage_form 1 / 2): the bits of form 0."""
    g = torch.Generator().manual_seed(m + c1)
    coarse, skip = torch.randn(ns, c1, generator=g).cuda(), torch.randn(m, c2, generator=g).cuda()
    idx = torch.randint(0, ns + 3, (m, 4), generator=g).cuda()  # (rows past the coarse level: zero rows)
    w = torch.zeros(c1 + c2, (n + 3) // 4 * 4)
    w[:, :n] = torch.randn(c1 + c2, n, generator=g) / (c1 + c2) ** 0.5
    w, bias, gamma, beta = w.cuda(), torch.randn(n, generator=g).cuda(), (torch.rand(n, generator=g) + 0.5).cuda(), torch.randn(n, generator=g).cuda()
    groups = 32 if n % 32 == 0 else 4
    run = lambda form: ops.decoder_stage(coarse, idx, skip, w, n, bias, gamma if norm else None, beta if norm else None, groups, act=ops.ACT_LEAKY, form=form)
    want = run(0)
    assert torch.equal(run(1), want) and torch.equal(run(2), want)
