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
    assert rre < 0.05 and rte < 5e-4, (rre, rte)
    r2 = eng.run(torch.from_numpy(rp).cuda(), torch.from_numpy(sp).cuda())
    assert np.array_equal(eng.transform(), T1) and r2.n_correspondences == n1  # run-to-run bit-reproducible
