"""GPU: the native engine (one call per pair) is bit-identical to the per-op Python mirror, which the
stage tests pin against the oracle; plus pose/NMS checks against the oracle directly."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx(oracle_native, golden_dir):
    from rdmnet_amd import collate, config, engine, model, weights
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    net = model.create_model(cfg).cuda()
    net.load_state_dict(state)
    eng = engine.Engine(cfg, state)
    eng.keep_taps(True)
    g = np.load(os.path.join(golden_dir, 'forward_small.npz'))
    return dict(cfg=cfg, state=state, net=net, eng=eng, rp=g['ref_points_in'], sp=g['src_points_in'], collate=collate)


def test_engine_equals_per_op_path_bit_for_bit(ctx):
    net, eng, cfg = ctx['net'], ctx['eng'], ctx['cfg']
    rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
    data = ctx['collate'].collate_pair(ctx['rp'], ctx['sp'], cfg)
    taps = {}
    out = net(data, taps)
    res = eng.run(rp, sp)
    for i in range(5):
        assert torch.equal(eng.tensor(f'points{i}'), data['points'][i])
        assert torch.equal(eng.tensor(f'neighbors{i}'), data['neighbors'][i])
    for name in taps:
        if name.startswith('encoder.'):
            assert torch.equal(eng.tensor(name), taps[name]), name
    n_c = int(data['lengths'][-1][0])
    assert torch.equal(eng.tensor('t1')[:n_c], taps['t1_ref'])
    assert torch.equal(eng.tensor('decoder'), taps['decoder'])
    assert torch.equal(eng.tensor('vote_xyz'), taps['vote_xyz'])
    assert torch.equal(eng.tensor('nms_mask')[:, 0], taps['nms_mask'])
    assert torch.equal(eng.tensor('feats_c')[:res.n_ref_nodes], out['ref_feats_c'])
    assert torch.equal(eng.tensor('ref_node_corr_indices')[:, 0], out['ref_node_corr_indices'])
    assert torch.equal(eng.tensor('matching_scores').reshape(-1, 129, 129), out['matching_scores'])
    rc, sc, cs = eng.corr()
    assert torch.equal(rc, out['ref_corr_points']) and torch.equal(sc, out['src_corr_points']) and torch.equal(cs, out['corr_scores'])
    assert np.array_equal(eng.transform(), out['estimated_transform'].cpu().numpy())
    assert torch.equal(eng.tensor('estimated_transform'), out['estimated_transform'])
    # the host half of the result (pinned buffer written by the run's last kernel) holds the same bits
    hr, hs, hc = eng.host_corr()
    assert res.n_host_correspondences == res.n_correspondences == hr.shape[0]
    assert np.array_equal(hr, rc.cpu().numpy()) and np.array_equal(hs, sc.cpu().numpy()) and np.array_equal(hc, cs.cpu().numpy())


def test_engine_matches_oracle_and_is_deterministic(ctx):
    from oracle import forward as ofw
    cfg, eng = ctx['cfg'], ctx['eng']
    rp, sp = ctx['rp'], ctx['sp']
    odata = ofw.pyramid(np.concatenate([rp, sp]), np.array([len(rp), len(sp)], np.int64), cfg)
    otaps = {}
    oout = ofw.forward(ofw.to_torch(ctx['state']), cfg, odata, otaps)
    r1 = eng.run(torch.from_numpy(rp).cuda(), torch.from_numpy(sp).cuda())
    T1, n1 = eng.transform(), r1.n_correspondences
    assert torch.equal(eng.tensor('nms_mask')[:, 0].cpu().bool(), otaps['nms_mask'])
    rre, rte = ofw.rre_rte(T1, oout['estimated_transform'].numpy())
    assert rre <= 1e-3 and rte <= 1e-5, (rre, rte)  # the north star's bound, end to end
    rc, sc, _ = eng.corr()
    assert torch.equal(rc.cpu(), oout['ref_corr_points']) and torch.equal(sc.cpu(), oout['src_corr_points'])
    r2 = eng.run(torch.from_numpy(rp).cuda(), torch.from_numpy(sp).cuda())
    assert np.array_equal(eng.transform(), T1) and r2.n_correspondences == n1  # run-to-run bit-reproducible


def test_engines_share_one_copy_of_the_weights(ctx):
    """rdm_engine_share_params: a second engine on the first one's prepared device parameters (what bench.py's in-flight
    engines and a module's per-stream engines do) gives the same bits, also after the first engine ran something else."""
    from rdmnet_amd import engine
    cfg, eng = ctx['cfg'], ctx['eng']
    rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
    n1 = eng.run(rp, sp).n_correspondences  # (the result structure is the engine's: read it before the next run)
    T1, c1 = eng.transform(), [x.copy() for x in eng.host_corr()]
    twin = engine.Engine(cfg, None, share_with=eng)
    eng.run(sp, rp)  # (the owner moves on)
    r2 = twin.run(rp, sp)
    assert r2.n_correspondences == n1 and np.array_equal(twin.transform(), T1)
    assert all(np.array_equal(a, b) for a, b in zip(twin.host_corr(), c1))


def test_shared_parameters_outlive_their_first_owner(ctx):
    """the prepared parameters are reference counted in the library.  The engine that uploaded them may be
    destroyed first (an LRU eviction does exactly that): its sharers keep working on the same bits, a sharer of a sharer
    too, and a configuration with another model shape is refused."""
    import copy
    import gc
    from rdmnet_amd import engine
    cfg = ctx['cfg']
    rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
    owner = engine.Engine(cfg, ctx['state'])
    n1 = owner.run(rp, sp).n_correspondences
    T1 = owner.transform()
    twin = engine.Engine(cfg, None, share_with=owner)
    grand = engine.Engine(cfg, None, share_with=twin)
    other = copy.deepcopy(cfg)
    other.model.num_points_in_patch = 64
    with pytest.raises(RuntimeError, match='different model shapes'):
        engine.Engine(other, None, share_with=owner)
    free0 = torch.cuda.mem_get_info()[0]
    del owner  # rdm_engine_destroy(owner): its 3 GiB arena goes, the parameters stay with twin / grand
    gc.collect()
    assert torch.cuda.mem_get_info()[0] - free0 >= (2 << 30)
    for e in (twin, grand):
        assert e.run(rp, sp).n_correspondences == n1 and np.array_equal(e.transform(), T1)
    del twin
    gc.collect()
    assert grand.run(rp, sp).n_correspondences == n1 and np.array_equal(grand.transform(), T1)


def test_pairs_in_flight_hint_changes_no_bit(ctx):
    """rdm_engine_set_pairs_in_flight(n >= 3) caps the tiled GEMM at two workgroups per CU (a scheduling hint for several pairs
    sharing the GPU): results are the same bits as without it."""
    eng = ctx['eng']
    rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
    eng.run(rp, sp)
    T1, c1, feats = eng.transform(), [x.copy() for x in eng.host_corr()], eng.tensor('decoder').clone()
    eng.set_pairs_in_flight(4)
    try:
        eng.run(rp, sp)
        assert np.array_equal(eng.transform(), T1) and all(np.array_equal(a, b) for a, b in zip(eng.host_corr(), c1))
        assert torch.equal(eng.tensor('decoder'), feats)
    finally:
        eng.set_pairs_in_flight(1)


def test_latency_mode_changes_no_bit(ctx):
    """rdm_engine_set_overlap: with one pair in flight the engine runs the first level's grid / search / encoder blocks beside the
    subsampling of the deeper levels, and the decoder beside the second transformer and the coarse matching, on a side stream of
    its own.  Same kernels on the same operands: every stage tensor, the pose and the correspondences are the bits of the serial
    run -- also when the caller's stream is still busy producing the inputs (the first half then stays serial), repeatedly."""
    eng = ctx['eng']
    rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
    names = ['points1', 'points4', 'neighbors0', 'subsampling0', 'neighbors4', 'encoder.encoder1_1', 'encoder.encoder1_2',
             'encoder.encoder2_1', 'encoder.encoder5_3', 't1', 'decoder', 'p2p_scores', 'vote_feats', 't2', 'feats_c',
             'matching_scores', 'search_flags']
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    try:
        with torch.cuda.stream(st):
            eng.set_overlap(0)
            eng.run(rp, sp)
            base = {k: eng.tensor(k).clone() for k in names}
            T0, c0 = eng.transform(), [x.copy() for x in eng.host_corr()]
            eng.set_overlap(2)
            for rep in range(3):
                if rep == 2:  # inputs still being produced on the stream when the run starts
                    big = torch.randn(4096, 4096, device='cuda')
                    for _ in range(4):
                        big = big @ big * 1e-3
                    r2, s2 = rp + 0 * big[0, 0], sp + 0 * big[0, 0]
                else:
                    r2, s2 = rp, sp
                eng.run(r2, s2)
                assert np.array_equal(eng.transform(), T0) and all(np.array_equal(a, b) for a, b in zip(eng.host_corr(), c0))
                for k in names:
                    assert torch.equal(eng.tensor(k), base[k]), k
        # the null stream: latency mode is skipped (every blocking stream serialises with it), results as ever
        eng.run(rp, sp)
        assert np.array_equal(eng.transform(), T0)
    finally:
        eng.set_overlap(1)


def test_full_size_pair_properties(ctx, golden_dir):
    """BASELINE-size workload (2 x 16k points): the engine equals the per-op mirror bit for bit, the pose
    is a proper rigid transform, correspondences are points of the fine level, neighbour tables are
    distance-sorted, and a second run reproduces everything exactly."""
    cfg, net, eng = ctx['cfg'], ctx['net'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    ref, src = z['ref0'], z['src0']
    rp, sp = torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda()
    res = eng.run(rp, sp)
    T = eng.transform()
    R = T[:3, :3].astype(np.float64)
    assert np.abs(R.T @ R - np.eye(3)).max() < 1e-5 and abs(np.linalg.det(R) - 1.0) < 1e-5
    assert np.array_equal(T[3], np.array([0, 0, 0, 1], np.float32))
    assert res.level_sizes[0] == len(ref) + len(src) and all(res.level_sizes[i] > res.level_sizes[i + 1] for i in range(4))
    rc, sc, cs = eng.corr()
    assert rc.shape[0] == res.n_correspondences > 0 and bool((cs > 0).all()) and bool((cs <= 1.0 + 1e-6).all())
    hr, hs, hc = eng.host_corr()
    assert np.array_equal(hr, rc.cpu().numpy()) and np.array_equal(hs, sc.cpu().numpy()) and np.array_equal(hc, cs.cpu().numpy())
    pts_f = eng.tensor('points1')
    n_ref_f = int(res.level_sizes[1])  # both clouds stacked; correspondences must be rows of it
    fine = {tuple(p) for p in pts_f.cpu().numpy().round(6).tolist()}
    assert all(tuple(p) in fine for p in rc.cpu().numpy().round(6).tolist()[:200])
    nb, p0 = eng.tensor('neighbors1').cpu(), pts_f.cpu()
    pad = torch.cat([p0, torch.full((1, 3), 1e6)])
    d = ((pad[nb] - p0[:, None]) ** 2).sum(-1)
    valid = nb < p0.shape[0]
    both = valid[:, 1:] & valid[:, :-1]
    assert bool((d[:, 1:][both] >= d[:, :-1][both]).all()) and bool((nb[:, 0] == torch.arange(p0.shape[0])).all())
    # per-op mirror, same inputs (a taps dictionary selects it; without one forward() is the native call)
    out = net(ctx['collate'].collate_pair(ref, src, cfg), {})
    assert np.array_equal(T, out['estimated_transform'].cpu().numpy())
    assert torch.equal(rc, out['ref_corr_points']) and torch.equal(cs, out['corr_scores'])
    # run-to-run
    res2 = eng.run(rp, sp)
    assert np.array_equal(eng.transform(), T) and res2.n_correspondences == rc.shape[0]


@pytest.mark.parametrize('seed', range(10))
def test_plain_engine_run_equals_per_op_mirror_on_random_scenes(ctx, seed):
    """Randomised end-to-end equality: a PLAIN rdm_engine_run (no stage tensors kept: the production form, which skips the
    up-sampling search nothing reads, shares another engine's parameters and hands its result over in pinned host memory)
    against the per-op Python mirror on the full collate -- lidar-like scenes of 2-25 k points per scan (a plane, boxes and
    clutter, the second scan moved and cropped), one in four with the GEMM residency hint.  Pose, correspondences and their
    scores must be the same bits."""
    from rdmnet_amd import engine
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(2000, 25000))
    extent = float(rng.uniform(15.0, 70.0))

    def scene(k):
        ground = np.c_[rng.uniform(-extent, extent, (k // 2, 2)), rng.normal(-1.7, 0.03, k // 2)]
        centres = rng.uniform(-extent, extent, (12, 2))
        walls = np.concatenate([np.c_[c + rng.uniform(-3, 3, (k // 30, 2)) * [1.0, 0.05], rng.uniform(-1.7, 4.0, k // 30)] for c in centres])
        clutter = np.c_[rng.uniform(-extent, extent, (k - k // 2 - 12 * (k // 30), 2)), rng.uniform(-1.7, 2.0, k - k // 2 - 12 * (k // 30))]
        return np.concatenate([ground, walls, clutter]).astype(np.float32)
    ref = scene(n)
    ang = rng.uniform(-0.3, 0.3)
    R = np.array([[np.cos(ang), -np.sin(ang), 0], [np.sin(ang), np.cos(ang), 0], [0, 0, 1]], np.float32)
    src = (ref[rng.random(n) < rng.uniform(0.6, 1.0)] - np.array([rng.uniform(2, 12), rng.uniform(-2, 2), 0], np.float32)) @ R
    src = (src + rng.normal(0, 0.02, src.shape)).astype(np.float32)
    cfg, net = ctx['cfg'], ctx['net']
    plain = engine.Engine(cfg, None, share_with=ctx['eng'])
    if seed % 4 == 3:
        plain.set_pairs_in_flight(4)
    res = plain.run(torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda())
    T, (hr, hs, hc) = plain.transform(), plain.host_corr()
    out = net(ctx['collate'].collate_pair(ref, src, cfg), {})
    assert np.array_equal(T, out['estimated_transform'].cpu().numpy())
    assert res.n_correspondences == out['corr_scores'].shape[0]
    assert np.array_equal(hr, out['ref_corr_points'].cpu().numpy()) and np.array_equal(hs, out['src_corr_points'].cpu().numpy())
    assert np.array_equal(hc, out['corr_scores'].cpu().numpy())


@pytest.mark.parametrize('n,scale', [(500, 20.0), (50, 5.0), (5, 1.0), (1, 1.0), (4000, 200.0)])
def test_engine_handles_degenerate_clouds(ctx, n, scale):
    """Tiny, single-point and extremely sparse clouds (every point its own voxel at all levels) run through the whole
    path: no hang, no capacity error, a finite pose, sizes that shrink monotonically."""
    rng = np.random.default_rng(n)
    a = torch.from_numpy((rng.uniform(-1, 1, (n, 3)) * np.array([scale, scale, 2.0])).astype(np.float32)).cuda()
    b = torch.from_numpy((rng.uniform(-1, 1, (max(n - 3, 1), 3)) * np.array([scale, scale, 2.0])).astype(np.float32)).cuda()
    res = ctx['eng'].run(a, b)
    assert np.isfinite(ctx['eng'].transform()).all()
    assert res.level_sizes[0] == a.shape[0] + b.shape[0]
    assert all(res.level_sizes[i] >= res.level_sizes[i + 1] >= 2 for i in range(4))
    assert res.n_ref_nodes >= 1 and res.n_src_nodes >= 1 and res.n_correspondences >= 0


def test_latency_mode_on_degenerate_clouds_and_after_an_error(ctx):
    """The side-stream mode on the inputs that bend the path -- single points, a few points, every point its own voxel -- gives
    the serial run's bits, and a run that fails between fork and join (a 4-point arena) leaves an engine that still works."""
    from rdmnet_amd import engine
    eng = ctx['eng']
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    try:
        with torch.cuda.stream(st):
            for n, scale in ((500, 20.0), (5, 1.0), (1, 1.0), (4000, 200.0)):
                rng = np.random.default_rng(n)
                a = torch.from_numpy((rng.uniform(-1, 1, (n, 3)) * np.array([scale, scale, 2.0])).astype(np.float32)).cuda()
                b = torch.from_numpy((rng.uniform(-1, 1, (max(n - 3, 1), 3)) * np.array([scale, scale, 2.0])).astype(np.float32)).cuda()
                st.synchronize()
                outs = []
                for mode in (0, 2):
                    eng.set_overlap(mode)
                    res = eng.run(a, b)
                    outs.append((eng.transform(), [x.copy() for x in eng.host_corr()], list(res.level_sizes), eng.tensor('decoder').clone()))
                assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][2] == outs[1][2]
                assert all(np.array_equal(x, y) for x, y in zip(outs[0][1], outs[1][1])) and torch.equal(outs[0][3], outs[1][3])
            # an arena too small for the pair, fixed by the caller: the run fails part-way (side stream busy), the next ones work
            small = engine.Engine(ctx['cfg'], None, arena_bytes=160 << 20, share_with=eng)
            small.set_overlap(2)
            rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
            st.synchronize()
            with pytest.raises(RuntimeError, match='arena'):
                small.run(rp, sp)
            with pytest.raises(RuntimeError, match='arena'):
                small.run(rp, sp)
            del small
            eng.run(rp, sp)
            T2 = eng.transform()
            eng.set_overlap(0)
            eng.run(rp, sp)
            assert np.array_equal(eng.transform(), T2)
    finally:
        eng.set_overlap(1)


def test_engine_deferred_large_buffer_pass_matches_per_op_searches(ctx):
    """6 000 points in a 6 x 6 x 4 m box: 300-500 points inside a level-0 search radius, more than the first pass's
    256-key buffer.  The engine defers the large-buffer pass of all 14 searches to one launch; the per-op path runs it
    right after each search: same neighbour tables."""
    rng = np.random.default_rng(5)
    box = np.array([3.0, 3.0, 2.0])
    a = (rng.uniform(-1, 1, (6000, 3)) * box).astype(np.float32)
    b = (rng.uniform(-1, 1, (5500, 3)) * box).astype(np.float32)
    data = ctx['collate'].collate_pair(a, b, ctx['cfg'])
    ctx['eng'].run(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    for i in range(5):
        assert torch.equal(ctx['eng'].tensor(f'points{i}'), data['points'][i])
        assert torch.equal(ctx['eng'].tensor(f'neighbors{i}'), data['neighbors'][i]), i
        if i < 4:
            assert torch.equal(ctx['eng'].tensor(f'subsampling{i}'), data['subsampling'][i]), i
            assert torch.equal(ctx['eng'].tensor(f'upsampling{i}'), data['upsampling'][i]), i
    # the case really overflows the small buffer
    from rdmnet_amd import ext
    p0, l0 = data['points'][0], data['lengths'][0]
    full = ext.radius_neighbors(p0, p0, l0, l0, float(ctx['cfg'].backbone.init_radius))
    assert int((full < p0.shape[0]).sum(1).max()) > 256


def test_neighbour_limits_beyond_128_slots(ctx, oracle_native):
    """Neighbour limits are whatever calibrate_neighbors_stack_mode returns (geotransformer/utils/data.py:195-220); round 2
    refused limits beyond the 128 slots the KPConv kernels stage in LDS (missing 5).  A dense cloud (hundreds
    of points inside a search radius) with limits of 150-200: the engine equals the per-op mirror bit for bit, and both
    follow the oracle run with the same limits (encoder taps 2e-5; the tables bit-exact)."""
    import copy
    from oracle import forward as ofw
    from rdmnet_amd import engine, model
    cfg = copy.deepcopy(ctx['cfg'])
    cfg.neighbor_limits = [200, 180, 160, 150, 150]
    rng = np.random.default_rng(9)
    box = np.array([10.0, 10.0, 2.0])
    a = (rng.uniform(-1, 1, (5000, 3)) * box).astype(np.float32)
    b = (rng.uniform(-1, 1, (4500, 3)) * box).astype(np.float32)
    odata = ofw.pyramid(np.concatenate([a, b]), np.array([len(a), len(b)], np.int64), cfg)
    assert max(int((t < p.shape[0]).sum(1).max()) for t, p in zip(odata['neighbors'], odata['points'])) > 128
    otaps = {}
    ofw.forward(ofw.to_torch(ctx['state']), cfg, odata, otaps)
    net = model.create_model(cfg).cuda()
    net.load_state_dict(ctx['state'])
    data = ctx['collate'].collate_pair(a, b, cfg, exact_shapes=True)
    for key in ('neighbors', 'subsampling', 'upsampling'):
        for x, y in zip(data[key], odata[key]):
            assert torch.equal(x.cpu(), y), key
    taps = {}
    out = net(data, taps)
    worst = 0.0
    for k in taps:
        if k.startswith('encoder.'):
            d, r = (taps[k].cpu().double() - otaps[k].double()).abs().max().item(), otaps[k].double().abs().max().item()
            worst = max(worst, d / r)
    assert worst <= 2e-5, worst
    eng = engine.Engine(cfg, ctx['state'])
    eng.keep_taps(True)
    eng.run(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    for k in taps:
        if k.startswith('encoder.'):
            assert torch.equal(eng.tensor(k), taps[k]), k
    assert np.array_equal(eng.transform(), out['estimated_transform'].cpu().numpy())


def test_engine_handles_neighbourhoods_beyond_every_buffer(ctx, oracle_native):
    """6 000 points in a 2.4 m cube put ~3 000 points into a level-0 search radius -- more than the kernels' largest
    (1024-key) buffer.  The reference returns every neighbour (radius_neighbors_cpu.cpp:36-64) and its callers keep the
    `limit` nearest (radius_search.py:24-26); the engine does the same through the multi-round select: its tables equal
    the per-op collate's and the oracle's (the reference's own C++ when oracle/_ref is present), and the pair runs
    through to a finite pose."""
    from oracle import forward as ofw
    rng = np.random.default_rng(0)
    a = rng.uniform(-1.2, 1.2, (6000, 3)).astype(np.float32)
    b = rng.uniform(-1.2, 1.2, (5000, 3)).astype(np.float32)
    res = ctx['eng'].run(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert np.isfinite(ctx['eng'].transform()).all() and res.level_sizes[0] == 11000
    data = ctx['collate'].collate_pair(a, b, ctx['cfg'])
    odata = ofw.pyramid(np.concatenate([a, b]), np.array([len(a), len(b)], np.int64), ctx['cfg'])
    for i in range(5):
        assert torch.equal(ctx['eng'].tensor(f'points{i}').cpu(), odata['points'][i])
        for key, tables in (('neighbors', data['neighbors']), ('subsampling', data['subsampling']), ('upsampling', data['upsampling'])):
            if key != 'neighbors' and i == 4:
                continue
            w = odata[key][i].shape[1]
            assert torch.equal(ctx['eng'].tensor(f'{key}{i}')[:, :w].cpu(), odata[key][i]), (key, i)
            assert torch.equal(tables[i][:, :w].cpu(), odata[key][i]), (key, i)
    from rdmnet_amd import ext
    p0, l0 = data['points'][0], data['lengths'][0]
    full = ext.radius_neighbors(p0, p0, l0, l0, float(ctx['cfg'].backbone.init_radius))
    assert int((full < p0.shape[0]).sum(1).max()) > 1024  # the case really exceeds the largest buffer


def test_engines_are_reentrant_across_threads_and_streams(ctx):
    """Four engines driven by four host threads on four streams (the bench's configuration) produce, for the same
    pair, bit-identical poses and correspondence counts: the library keeps no global state."""
    import threading
    from rdmnet_amd import engine
    rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
    ref_T, ref_n = None, None
    ctx['eng'].run(rp, sp)
    ref_T, ref_n = ctx['eng'].transform(), ctx['eng'].result.n_correspondences
    engines = [engine.Engine(ctx['cfg'], ctx['state']) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    out, errs = [None] * 4, []

    def work(k):
        try:
            with torch.cuda.stream(streams[k]):
                res = []
                for _ in range(6):
                    r = engines[k].run(rp, sp)
                    res.append((engines[k].transform(), r.n_correspondences))
                out[k] = res
        except BaseException as e:
            errs.append(e)
    torch.cuda.synchronize()
    ts = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    for res in out:
        for T, n in res:
            assert n == ref_n and np.array_equal(T, ref_T)


@pytest.mark.parametrize('exact_shapes', [False, True])
def test_model_forward_is_the_native_call_and_equals_engine_and_per_op_paths(ctx, golden_dir, exact_shapes):
    """The drop-in operator: model(data_dict) (experiments/model_infer.py:109-354) runs rdm_engine_forward on the
    collate's tables -- every one of the reference's 31 output keys equals the per-op mirror's bit for bit, and the pose /
    correspondences equal rdm_engine_run's (which builds the same tables itself).  exact_shapes=True hands over index
    tensors sliced to the reference's exact widths (non-contiguous views, as the reference's radius_search returns)."""
    cfg, net, eng = ctx['cfg'], ctx['net'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    for ref, src in ((ctx['rp'], ctx['sp']), (z['ref1'], z['src1'])):
        data = ctx['collate'].collate_pair(ref, src, cfg, exact_shapes=exact_shapes)
        if exact_shapes:
            assert not data['neighbors'][0].is_contiguous() or data['neighbors'][0].shape[1] == cfg.neighbor_limits[0]
        out = net(data)
        ref_out = net(data, {})
        assert set(out.keys()) == set(ref_out.keys()) and len(out) == 31
        for k in out:
            assert out[k].dtype == ref_out[k].dtype and out[k].shape == ref_out[k].shape, k
            assert torch.equal(out[k], ref_out[k]), k
        eng.run(torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda())
        assert np.array_equal(eng.transform(), out['estimated_transform'].cpu().numpy())
        rc, sc, cs = eng.corr()
        assert torch.equal(rc, out['ref_corr_points']) and torch.equal(cs, out['corr_scores'])


def test_model_forward_in_latency_mode_changes_no_bit(ctx):
    """model(data_dict) on a stream of the caller's (not the null stream) with one pair in flight runs its decoder on the engine's
    side stream (rdm_engine_set_overlap, the default): all 31 outputs are the bits of the null-stream call."""
    cfg, net = ctx['cfg'], ctx['net']
    data = ctx['collate'].collate_pair(ctx['rp'], ctx['sp'], cfg)
    base = {k: v.clone() for k, v in net(data).items()}
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(2):
            out = net(data)
            st.synchronize()
            assert set(out) == set(base)
            for k in out:
                assert torch.equal(out[k], base[k]), k


def test_model_is_a_torch_module_with_the_reference_checkpoint_layout(ctx):
    """nn.Module surface the reference's harness uses (engine/base_tester.py:97-113): strict load_state_dict with the 497
    checkpoint keys, .parameters(), .to(), .eval(); a wrong shape or a missing key raises RuntimeError."""
    from rdmnet_amd import model, weights
    cfg = ctx['cfg']
    net = model.create_model(cfg)
    assert isinstance(net, torch.nn.Module) and not net.training or net.eval() is net
    sd = net.state_dict()
    assert list(sd.keys()) == list(weights.schema(cfg).keys()) and len(sd) == 497
    n_par, n_buf = sum(p.numel() for p in net.parameters()), sum(b.numel() for b in net.buffers())
    assert n_buf == 14 * 15 * 3 and n_par + n_buf == sum(int(np.prod(s)) for s in weights.schema(cfg).values())
    assert all(p.requires_grad for p in net.parameters())  # as the reference's module (DDP needs one; forward runs under no_grad)
    assert net.load_state_dict(ctx['state'], strict=True) is not None
    bad = dict(ctx['state'])
    bad.pop('optimal_transport.alpha')
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    bad = dict(ctx['state'])
    bad['proj_n2p_score.weight'] = np.zeros((2, 256), np.float32)
    with pytest.raises(RuntimeError):
        net.load_state_dict(bad, strict=True)
    net = net.to('cuda').eval()
    assert next(net.parameters()).is_cuda and net.device.type == 'cuda'
    out = net(ctx['collate'].collate_pair(ctx['rp'], ctx['sp'], cfg))
    assert out['estimated_transform'].shape == (4, 4)


def test_native_collate_equals_per_op_collate_and_feeds_the_model(ctx, golden_dir):
    """registration_collate_fn_stack_mode(..., engine=model.engine()) -- the reference's collate as ONE native call
    (rdm_engine_collate) -- returns the same data_dict as the 17-launch Python collate, tensor for tensor (points, lengths,
    the 13 index tables up to their effective widths, features), in fresh tensors that survive further engine runs; the model
    on it equals rdm_engine_run."""
    cfg, net, eng, collate = ctx['cfg'], ctx['net'], ctx['eng'], ctx['collate']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    b = cfg.backbone
    for ref, src in ((ctx['rp'], ctx['sp']), (z['ref2'], z['src2'])):
        item = {'seq_id': 3, 'ref_frame': 10, 'src_frame': 11, 'ref_points': ref, 'src_points': src,
                'ref_feats': np.ones((len(ref), 1), np.float32), 'src_feats': np.ones((len(src), 1), np.float32)}
        args = ([item], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits)
        nat = collate.registration_collate_fn_stack_mode(*args, engine=net.engine(), exact_shapes=True)
        ref_d = collate.registration_collate_fn_stack_mode(*args, exact_shapes=True)
        assert nat['seq_id'] == 3 and nat['batch_size'] == 1 and torch.equal(nat['features'], ref_d['features'])
        for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
            assert len(nat[key]) == len(ref_d[key])
            for a, c in zip(nat[key], ref_d[key]):
                assert a.dtype == c.dtype and torch.equal(a, c), key
        net.engine().run(torch.from_numpy(src).cuda(), torch.from_numpy(ref).cuda())  # overwrite the arena: the dict must not care
        out = net(nat)
        eng.run(torch.from_numpy(ref).cuda(), torch.from_numpy(src).cuda())
        assert np.array_equal(eng.transform(), out['estimated_transform'].cpu().numpy())


def test_plain_engine_runs_and_int64_per_op_calls_interleave_on_one_thread(ctx):
    """VERDICT r4 (next 7): the library keeps no mode between calls -- the index width of a neighbour table and the GroupNorm
    form are arguments of the kernels' launches, not thread-local state (round 4 held them in two thread_local switches that
    the engine set around its own calls).  A plain engine run (int32 tables inside) and per-op calls on the reference's
    int64 tables, in both GroupNorm forms, alternate on ONE thread and every call returns what it returns on its own."""
    from rdmnet_amd import engine, ops
    cfg, eng = ctx['cfg'], ctx['eng']
    rp, sp = torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()
    data = ctx['collate'].collate_pair(ctx['rp'], ctx['sp'], cfg)
    plain = engine.Engine(cfg, None, share_with=eng)  # keep_taps off: the run keeps its tables in 32 bits
    g = torch.Generator().manual_seed(5)
    n1 = data['points'][1].shape[0]
    x = torch.randn(n1, 128, generator=g).cuda()
    gam, bet = torch.randn(128, generator=g).cuda(), torch.randn(128, generator=g).cuda()
    sub = data['subsampling'][1]  # int64 [n2, limit], pad = n1

    def per_op():
        return (ops.gather_max(x, sub), ops.group_norm(x, gam, bet, 32, act=ops.ACT_LEAKY, form=1),
                ops.group_norm(x[:2048], gam, bet, 32, act=ops.ACT_LEAKY), ops.group_norm(x[:2048], gam, bet, 32, act=ops.ACT_LEAKY, form=1),
                ctx['collate'].collate_pair(ctx['rp'], ctx['sp'], cfg)['neighbors'][2])
    want = per_op()
    assert torch.equal(want[0], torch.cat([x, torch.zeros(1, 128, device='cuda')])[sub].max(1)[0])  # int64 rows were read as int64
    assert torch.equal(want[2], want[3])  # (one-launch and three-launch GroupNorm: the same bits)
    assert torch.equal(want[4], data['neighbors'][2])
    eng.run(rp, sp)
    T, corr = eng.transform(), [c.clone() for c in eng.corr()]
    for _ in range(2):
        r = plain.run(rp, sp)
        assert np.array_equal(plain.transform(), T) and r.n_correspondences == corr[0].shape[0]
        got = per_op()
        assert all(torch.equal(a, b) for a, b in zip(got, want))
    assert all(torch.equal(a, b) for a, b in zip(plain.corr(), corr))


def test_batched_collate_gives_every_pair_the_bits_of_its_own_run(ctx, golden_dir):
    """Round 5: rdm_engine_collate_batch builds the pyramids of several pairs with ONE sequence of launches (2 B clouds per
    subsampling launch, (pair, level) grid items, 16 searches per query launch, one read-back of all level sizes);
    rdm_engine_forward_batched(k) runs pair k's forward on them.  Pairs of five very different sizes in one batch: pose,
    correspondences (points and scores), counters and level sizes of every pair equal those of rdm_engine_run on the pair alone,
    bit for bit; `run` recognises the next prepared pair by its tensors and drops the batch when another pair comes."""
    from rdmnet_amd import engine
    cfg, eng = ctx['cfg'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    sc = np.load(os.path.join(golden_dir, 'scans.npz'))

    def crop(p, r):
        return p[np.linalg.norm(p[:, :2], axis=1) < r]
    clouds = [(ctx['rp'], ctx['sp']), (z['ref0'], z['src0']), (crop(sc['s000000'], 14.0), crop(sc['s000004'], 12.0)),
              (z['ref1'], z['src1']), (sc['s000000'], sc['s000007'])]
    pairs = [(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) for a, b in clouds]
    plain = engine.Engine(cfg, None, share_with=eng)  # keep_taps off

    def snapshot(e, res):
        return (e.transform().copy(), [c.clone() for c in e.corr()], int(res.n_correspondences), int(res.n_ref_nodes), int(res.n_src_nodes),
                int(res.n_node_correspondences), [int(x) for x in res.level_sizes], [int(x) for x in res.level_ref_sizes])
    want = [snapshot(plain, plain.run(r, s)) for r, s in pairs]

    def same(a, b):
        return (np.array_equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1])) and a[2:] == b[2:])
    batched = engine.Engine(cfg, None, share_with=eng)
    for lo, hi in ((0, 5), (1, 3), (4, 5)):  # batches of 5, 2 and 1 pairs on one engine, one after the other
        assert batched.collate_batch(pairs[lo:hi]) == hi - lo
        for k in range(lo, hi):
            assert same(snapshot(batched, batched.run(*pairs[k])), want[k]), (lo, hi, k)
    # forwards of a batch in any order, and twice
    batched.collate_batch(pairs[:3])
    for k in (2, 0, 2, 1):
        assert same(snapshot(batched, batched.forward_batched(k)), want[k]), k
    # another pair in between drops the batch: run() collates it itself, the prepared pair after it as well
    batched.collate_batch(pairs[:2])
    assert same(snapshot(batched, batched.run(*pairs[3])), want[3])
    assert same(snapshot(batched, batched.run(*pairs[0])), want[0])
    with pytest.raises(RuntimeError):
        batched.forward_batched(0)
    # keeping stage tensors builds the reference's full tables pair by pair: refused
    batched.keep_taps(True)
    with pytest.raises(RuntimeError):
        batched.collate_batch(pairs[:2])


def test_lockstep_runs_give_every_pair_the_bits_of_its_own_run(ctx, golden_dir):
    """Round 5 (experimental): rdm_engine_run_lockstep runs several pairs on as many engines on ONE stream -- stackful contexts of
    the calling thread, launches of converted kernels recorded and issued as one grouped launch per kernel, one host wait per
    read-back for the whole group.  Pairs of very different sizes (so that their grids, and the points at which they wait, differ):
    pose, correspondences and counters of every pair equal rdm_engine_run on the pair alone, bit for bit; group sizes 1 .. 5."""
    from rdmnet_amd import engine
    cfg, eng = ctx['cfg'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    sc = np.load(os.path.join(golden_dir, 'scans.npz'))

    def crop(p, r):
        return p[np.linalg.norm(p[:, :2], axis=1) < r]
    clouds = [(ctx['rp'], ctx['sp']), (z['ref0'], z['src0']), (crop(sc['s000000'], 14.0), crop(sc['s000004'], 12.0)),
              (z['ref1'], z['src1']), (sc['s000000'], sc['s000007'])]
    pairs = [(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()) for a, b in clouds]
    engines = [engine.Engine(cfg, None, share_with=eng) for _ in range(5)]

    def snapshot(e, res):
        return (e.transform().copy(), [c.clone() for c in e.corr()], int(res.n_correspondences), int(res.n_ref_nodes), int(res.n_src_nodes),
                int(res.n_node_correspondences), [int(x) for x in res.level_sizes])
    want = [snapshot(engines[0], engines[0].run(r, s)) for r, s in pairs]

    def same(a, b):
        return np.array_equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1])) and a[2:] == b[2:]
    with torch.cuda.stream(torch.cuda.Stream()):
        # (collates of the group as one launch sequence on the first engine -- rdm_engine_collate_batch -- or pair by pair in lock step)
        for collated, lo, hi in ((True, 0, 5), (False, 1, 3), (True, 3, 4), (False, 0, 4), (True, 2, 5)):
            res = engine.Engine.run_lockstep(engines, pairs[lo:hi], collate_batched=collated)
            for k in range(lo, hi):
                assert same(snapshot(engines[k - lo], res[k - lo]), want[k]), (collated, lo, hi, k)
        # the same pair several times in a group: every launch of the runs is then grouped, the size-dependent ones too
        for collated in (True, False):
            res = engine.Engine.run_lockstep(engines, [pairs[2], pairs[2], pairs[0], pairs[2]], collate_batched=collated)
            for k, i in enumerate((2, 2, 0, 2)):
                assert same(snapshot(engines[k], res[k]), want[i]), (collated, k)
        # an engine of a group runs alone again afterwards
        assert same(snapshot(engines[2], engines[2].run(*pairs[4])), want[4])


def test_lockstep_pairs_that_exhaust_their_arenas_are_rerun_and_keep_their_bits(ctx, golden_dir):
    """A pair of a lock-step group whose engine runs out of arena ends its run early (the others go on), is run again on its own with
    a grown arena (as rdm_engine_run does) and returns the bits of its own run; the first engine also grows while it collates the
    group.  Engines start with 160 MB (rdm_engine_reserve: growable), a 2 x 16 k-point pair needs several times that."""
    from rdmnet_amd import engine
    cfg, eng = ctx['cfg'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    pairs = [(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda()), (torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()),
             (torch.from_numpy(z['ref1']).cuda(), torch.from_numpy(z['src1']).cuda())]

    def snapshot(e, res):
        return (e.transform().copy(), [c.clone() for c in e.corr()], int(res.n_correspondences), [int(x) for x in res.level_sizes])
    want = [snapshot(eng, eng.run(r, s)) for r, s in pairs]
    for collated in (True, False):
        engines = [engine.Engine(cfg, None, share_with=eng) for _ in range(3)]
        for e in engines:
            e.reserve(160 << 20)
        with torch.cuda.stream(torch.cuda.Stream()):
            for _ in range(2):  # (the second group finds the arenas grown)
                res = engine.Engine.run_lockstep(engines, pairs, collate_batched=collated)
                for k in range(3):
                    got = snapshot(engines[k], res[k])
                    assert np.array_equal(got[0], want[k][0]) and all(torch.equal(a, b) for a, b in zip(got[1], want[k][1])) and got[2:] == want[k][2:], (collated, k)
                assert all(int(r.arena_used) > (160 << 20) for r in res)  # (every engine had to grow)


@pytest.mark.parametrize('variant', ['bf16_attention', 'vote_off'])
def test_lockstep_groups_of_the_configuration_variants_keep_their_bits(ctx, golden_dir, variant):
    """BASELINE configs[3] (bf16 attention: another attention kernel instantiation in the grouped launches) and configs[4]'s
    vote-off mode (another launch sequence): a lock-step group returns what rdm_engine_run returns for each pair."""
    import copy
    from rdmnet_amd import engine
    cfg = copy.deepcopy(ctx['cfg'])
    if variant == 'bf16_attention':
        cfg.thdroformer.attention_bf16 = True
    else:
        cfg.Vote.inference_use_vote = False
    first = engine.Engine(cfg, ctx['state'])
    engines = [first] + [engine.Engine(cfg, None, share_with=first) for _ in range(2)]
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    pairs = [(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda()), (torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()),
             (torch.from_numpy(z['ref1']).cuda(), torch.from_numpy(z['src1']).cuda())]

    def snapshot(e, res):
        return (e.transform().copy(), [c.clone() for c in e.corr()], int(res.n_correspondences), int(res.n_ref_nodes), int(res.n_src_nodes))
    want = [snapshot(engines[2], engines[2].run(r, s)) for r, s in pairs]
    with torch.cuda.stream(torch.cuda.Stream()):
        res = engine.Engine.run_lockstep(engines, pairs)
        for k in range(3):
            got = snapshot(engines[k], res[k])
            assert np.array_equal(got[0], want[k][0]) and all(torch.equal(a, b) for a, b in zip(got[1], want[k][1])) and got[2:] == want[k][2:], k


def test_model_on_a_list_of_data_dicts_runs_in_lock_step_and_equals_the_batch_1_calls(ctx, golden_dir):
    """Round 6: the drop-in operator API in lock step -- `model([data_dict, ...])` =
    rdm_engine_forward_lockstep on the callers' data_dicts (experiments/model_infer.py:109-354 per pair) -- returns, for every
    pair, the 31-key output_dict of `model(data_dict)` on it alone, `torch.equal` key by key; pairs of very different sizes,
    groups of 1 .. 4, data_dicts from the native collate and from the per-op collate (other table widths)."""
    net, cfg, collate = ctx['net'], ctx['cfg'], ctx['collate']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    sc = np.load(os.path.join(golden_dir, 'scans.npz'))

    def crop(p, r):
        return p[np.linalg.norm(p[:, :2], axis=1) < r]
    clouds = [(ctx['rp'], ctx['sp']), (z['ref0'], z['src0']), (crop(sc['s000000'], 14.0), crop(sc['s000004'], 12.0)), (z['ref1'], z['src1'])]
    with torch.cuda.stream(torch.cuda.Stream()):
        dicts = [collate.collate_pair(r, s, cfg, exact_shapes=(k % 2 == 0)) for k, (r, s) in enumerate(clouds)]
        want = [net(d) for d in dicts]
        assert all(len(w) == 31 for w in want)
        for lo, hi in ((0, 4), (1, 3), (2, 3), (0, 2)):
            got = net(dicts[lo:hi])
            assert isinstance(got, list) and len(got) == hi - lo
            for k, g in zip(range(lo, hi), got):
                assert set(g) == set(want[k])
                for key in g:
                    assert g[key].dtype == want[k][key].dtype and torch.equal(g[key], want[k][key]), (lo, hi, k, key)
        # batch 1 stays batch 1 (a dict in, a dict out), and still equals itself after the groups
        again = net(dicts[1])
        assert isinstance(again, dict) and all(torch.equal(again[key], want[1][key]) for key in again)
        with pytest.raises(ValueError):
            net(dicts * 3)  # 12 > 8 per group


def test_lockstep_groups_that_keep_their_stage_tensors_hold_every_pairs_tensors(ctx, golden_dir):
    """Round 6: engines that keep their stage tensors run in lock step too (each collates its own pair
    inside the group -- the reference's full tables): afterwards every engine holds the tensors of ITS pair, equal to those of a
    run on the pair alone."""
    from rdmnet_amd import engine
    cfg, eng = ctx['cfg'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    pairs = [(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda()), (torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()),
             (torch.from_numpy(z['ref1']).cuda(), torch.from_numpy(z['src1']).cuda())]
    names = ['points0', 'points4', 'neighbors0', 'upsampling0', 'subsampling3', 'encoder.encoder1_1', 'encoder.encoder4_3', 't1', 'decoder',
             'vote_xyz', 'nms_mask', 'feats_c', 'matching_scores', 'ref_corr_points', 'corr_scores', 'estimated_transform']
    want = []
    for r, s in pairs:
        eng.run(r, s)
        want.append({n: eng.tensor(n).clone() for n in names})
    engines = [engine.Engine(cfg, None, share_with=eng) for _ in range(3)]
    for e in engines:
        e.keep_taps(True)
    with torch.cuda.stream(torch.cuda.Stream()):
        for _ in range(2):
            engine.Engine.run_lockstep(engines, pairs)
            for k, e in enumerate(engines):
                for n in names:
                    assert torch.equal(e.tensor(n), want[k][n]), (k, n)


def test_a_lock_step_result_answers_one_run_of_exactly_its_tensors(ctx, golden_dir):
    """`Engine.run` right after a lock-step group returns the pair's result without running again -- but only for the
    very tensors the group ran on, unchanged (same memory AND same version counter), and only once: an in-place refill of the same
    buffers, another pair, or a second call run the pair."""
    from rdmnet_amd import engine
    cfg, eng = ctx['cfg'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    a = (torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda())
    b = (torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda())
    engines = [engine.Engine(cfg, None, share_with=eng) for _ in range(2)]
    Ta = engines[0].run(*a) and engines[0].transform().copy()
    Tb = engines[0].run(*b) and engines[0].transform().copy()
    assert not np.array_equal(Ta, Tb)
    calls = []
    real = engines[0].L.rdm_engine_run

    class Spy:  # counts the native runs of engines[0]
        def __getattr__(self, name):
            if name == 'rdm_engine_run':
                return lambda *args: (calls.append(1), real(*args))[1]
            return getattr(engines[0].__dict__['_L_real'], name)
    engines[0].__dict__['_L_real'] = engines[0].L
    engines[0].L = Spy()
    try:
        with torch.cuda.stream(torch.cuda.Stream()):
            engine.Engine.run_lockstep(engines, [a, b])
            engines[0].run(*a)                      # picked up: no native run
            assert not calls and np.array_equal(engines[0].transform(), Ta)
            engines[0].run(*a)                      # a second call runs
            assert len(calls) == 1 and np.array_equal(engines[0].transform(), Ta)
            # the same buffers refilled in place with another pair of the same size: must run, and return the new pair's pose
            buf = (a[0].clone(), a[1].clone())
            engine.Engine.run_lockstep(engines, [buf, b])
            n0, n1 = min(len(buf[0]), len(b[0])), min(len(buf[1]), len(b[1]))
            buf[0][:n0].copy_(b[0][:n0])
            buf[1][:n1].copy_(b[1][:n1])
            engines[0].run(*buf)
            assert len(calls) == 2 and not np.array_equal(engines[0].transform(), Ta)
            # clear_pending: what the pipeline does when a job's fn does not pick its result up
            engine.Engine.run_lockstep(engines, [a, b])
            engines[0].clear_pending()
            engines[0].run(*a)
            assert len(calls) == 3
    finally:
        engines[0].L = engines[0].__dict__.pop('_L_real')


def test_collates_in_lock_step_equal_the_one_pair_collates(ctx, golden_dir):
    """Round 6: `registration_collate_lockstep` = the drop-in collate of several pairs as one lock-step group
    (rdm_engine_collate_lockstep): every data_dict equals `registration_collate_fn_stack_mode([item], ..., engine=e)` on its pair
    -- all 13 tables at the reference's widths, points, lengths, status words -- and the dicts feed `model([...])`."""
    net, cfg, collate = ctx['net'], ctx['cfg'], ctx['collate']
    b = cfg.backbone
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    clouds = [(ctx['rp'], ctx['sp']), (z['ref0'], z['src0']), (z['ref1'], z['src1'])]
    items = [{'ref_points': r, 'src_points': s, 'ref_feats': np.ones((len(r), 1), np.float32), 'src_feats': np.ones((len(s), 1), np.float32),
              'seq_id': k} for k, (r, s) in enumerate(clouds)]
    with torch.cuda.stream(torch.cuda.Stream()):
        want = [collate.registration_collate_fn_stack_mode([it], b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, engine=net.engine())
                for it in items]
        got = collate.registration_collate_lockstep(items, b.num_stages, b.init_voxel_size, b.init_radius, cfg.neighbor_limits, net.engine_group(3))
        assert len(got) == 3
        for g, w in zip(got, want):
            assert g['seq_id'] == w['seq_id'] and g['_level_ref_sizes'] == w['_level_ref_sizes']
            assert torch.equal(g['features'], w['features']) and torch.equal(g['_flags'], w['_flags'])
            for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
                assert len(g[key]) == len(w[key]) and all(torch.equal(a, c) for a, c in zip(g[key], w[key])), key
        for g in got:
            g['testing'] = True
        outs, ref = net(got), [net(w) for w in want]
        for o, r in zip(outs, ref):
            assert all(torch.equal(o[key], r[key]) for key in r)


def test_lockstep_collates_and_forwards_that_exhaust_their_arenas_are_rerun(ctx, golden_dir):
    """The arena-growth path of the round-6 lock-step entry points: engines that start with 160 MB (rdm_engine_reserve: growable)
    cannot hold a 2 x 16 k-point pair; `rdm_engine_collate_lockstep` / `rdm_engine_forward_lockstep` end such a pair's run early,
    run it again on its own with a grown arena, and return what the one-pair calls return."""
    from rdmnet_amd import engine
    cfg, eng = ctx['cfg'], ctx['eng']
    z = np.load(os.path.join(golden_dir, 'synthetic_pairs.npz'))
    pairs = [(torch.from_numpy(z['ref0']).cuda(), torch.from_numpy(z['src0']).cuda()), (torch.from_numpy(ctx['rp']).cuda(), torch.from_numpy(ctx['sp']).cuda()),
             (torch.from_numpy(z['ref1']).cuda(), torch.from_numpy(z['src1']).cuda())]
    names = ['decoder', 'feats_c', 'matching_scores', 'ref_corr_points', 'src_corr_points', 'corr_scores', 'estimated_transform']
    want_d, want_t = [], []
    for r, s in pairs:
        d = eng.collate(r, s)
        eng.forward(d)
        want_d.append(d)
        want_t.append({n: eng.tensor(n).clone() for n in names})
    engines = [engine.Engine(cfg, None, share_with=eng) for _ in range(3)]
    for e in engines:
        e.keep_taps(True)
        e.reserve(160 << 20)
    with torch.cuda.stream(torch.cuda.Stream()):
        dicts = engine.Engine.collate_lockstep(engines, pairs)
        for d, w in zip(dicts, want_d):
            for key in ('points', 'lengths', 'neighbors', 'subsampling', 'upsampling'):
                assert all(torch.equal(a, b) for a, b in zip(d[key], w[key])), key
        for e in engines:
            e.reserve(160 << 20)  # (the collates grew the arenas: start small again for the forwards)
        res = engine.Engine.forward_lockstep(engines, dicts)
        assert any(int(r.arena_used) > (160 << 20) for r in res)
        for k, e in enumerate(engines):
            for n in names:
                assert torch.equal(e.tensor(n), want_t[k][n]), (k, n)
