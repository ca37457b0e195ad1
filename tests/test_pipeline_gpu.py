"""GPU: several pairs in flight (rdmnet_amd.pipeline.PairPipeline, what `python -m rdmnet_amd.infer` and bench.py run on)
give the bits of a serial run, in dataset order."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def setup(golden_dir):
    from rdmnet_amd import config, weights
    cfg = config.make_cfg()
    state = weights.synthetic_state_dict(cfg, seed=0)
    scans = np.load(os.path.join(golden_dir, 'scans.npz'))
    a, b, c = scans['s000000'], scans['s000004'], scans['s000007']

    def crop(p, r):
        return np.ascontiguousarray(p[np.linalg.norm(p[:, :2], axis=1) < r])
    # five distinct pairs of different sizes (their stages take different times, so completion order differs from input order)
    distinct = [(crop(a, 10.0), crop(b, 10.0)), (crop(a, 16.0), crop(b, 16.0)), (crop(a, 12.0), crop(c, 12.0)),
                (crop(b, 9.0), crop(a, 9.0)), (crop(a, 20.0), crop(b, 20.0))]
    return cfg, state, distinct


def test_pairs_in_flight_equal_the_serial_run_bit_for_bit_in_dataset_order(setup):
    from rdmnet_amd import pipeline
    cfg, state, distinct = setup
    order = [0, 1, 2, 3, 4, 4, 0, 3, 1, 2, 0, 0, 4, 2, 3, 1, 1, 4]
    dev = [(torch.from_numpy(r).cuda(), torch.from_numpy(s).cuda()) for r, s in distinct]
    serial = pipeline.PairPipeline(cfg, state, pairs_in_flight=1)
    want = serial.run_pairs([dev[i] for i in order])
    flight = pipeline.PairPipeline(cfg, None, pairs_in_flight=4, engines=[serial.engines[0]])  # (shares the weights)
    assert len(flight.engines) == 4 and len({id(s) for s in flight.streams}) == 4
    for _ in range(2):
        got = flight.run_pairs([dev[i] for i in order])
        assert len(got) == len(want)
        for g, w in zip(got, want):
            assert np.array_equal(g.transform, w.transform)
            assert g.n_correspondences == w.n_correspondences and g.level_sizes == w.level_sizes
            assert np.array_equal(g.ref_corr_points, w.ref_corr_points) and np.array_equal(g.src_corr_points, w.src_corr_points)
            assert np.array_equal(g.corr_scores, w.corr_scores)
    # pairs of one size gave one result (the engines are deterministic and independent)
    for i in set(order):
        rows = [g for g, k in zip(got, order) if k == i]
        assert all(np.array_equal(rows[0].transform, r.transform) for r in rows[1:])
    # round 5: the workers collate several drawn pairs with one sequence of launches (collate_batch > 1: batches of pairs of
    # different sizes, the last rounds shrunk) -- still the serial run's bits, in dataset order
    for cb in (3, 4):
        batched = pipeline.PairPipeline(cfg, None, pairs_in_flight=4, engines=[serial.engines[0]], collate_batch=cb)
        got = batched.run_pairs([dev[i] for i in order])
        assert batched.last_stats['collate_batches'] >= 2 and len(got) == len(want)
        for g, w in zip(got, want):
            assert np.array_equal(g.transform, w.transform) and g.level_sizes == w.level_sizes
            assert np.array_equal(g.ref_corr_points, w.ref_corr_points) and np.array_equal(g.src_corr_points, w.src_corr_points)
            assert np.array_equal(g.corr_scores, w.corr_scores)
    # round 5: lock-step groups (the default of a pipeline with its own engines): a worker runs the pairs it drew -- of different
    # sizes -- as one group on its stream, identical kernels of the group's pairs as one grouped launch; still the serial run's
    # bits, in dataset order
    for n, lb in ((2, 4), (3, 3), (4, 2)):
        grouped = pipeline.PairPipeline(cfg, None, pairs_in_flight=n, engines=[serial.engines[0]], lockstep=lb)
        # (round 6: a pipeline builds one engine per worker and completes its lock-step groups on their first draw)
        assert len(grouped.groups) == n and all(len(g) == 1 for g in grouped.groups) and grouped.engines[0] is serial.engines[0]
        for _ in range(2):
            got = grouped.run_pairs([dev[i] for i in order])
            assert all(len(g) == lb for g in grouped.groups)
            assert grouped.last_stats['lockstep_groups'] >= 2 and len(got) == len(want)
            for g, w in zip(got, want):
                assert np.array_equal(g.transform, w.transform) and g.level_sizes == w.level_sizes
                assert g.n_correspondences == w.n_correspondences
                assert np.array_equal(g.ref_corr_points, w.ref_corr_points) and np.array_equal(g.src_corr_points, w.src_corr_points)
                assert np.array_equal(g.corr_scores, w.corr_scores)
    assert pipeline.PairPipeline(cfg, state, pairs_in_flight=2).lockstep == pipeline.DEFAULT_LOCKSTEP
    assert pipeline.PairPipeline(cfg, state, pairs_in_flight=1).lockstep == 1


def test_tester_with_pairs_in_flight_writes_the_serial_outputs(setup, tmp_path):
    """`Tester.run` (python -m rdmnet_amd.infer) with four pairs in flight: records, pose lines (dataset order) and the
    per-pair .npz files equal the one-pair-at-a-time run's."""
    from rdmnet_amd import dataset, infer
    cfg, state, distinct = setup
    pairs = [distinct[i] + (np.eye(4),) for i in (0, 1, 2, 3, 4, 1, 0, 2)]
    outs = {}
    for n in (1, 4):
        d = tmp_path / f'flight{n}'
        t = infer.Tester(cfg, state, str(d), ransac=False, pairs_in_flight=n)
        recs = t.run(dataset.PairStager(dataset.ArrayPairDataset(pairs), depth=2 * n))
        outs[n] = (recs, open(d / '00_pose').read(), t.summary.lines())
        assert [r['ref_frame'] for r in recs] == [2 * i for i in range(len(pairs))]
        if n == 4:  # round 6: runs that keep their stage tensors (the .npz outputs) are lock-step groups too
            assert t.pipeline.lockstep == 4 and t.pipeline.keep_taps and t.pipeline.last_stats['lockstep_groups'] >= 1
    (r1, pose1, rep1), (r4, pose4, rep4) = outs[1], outs[4]
    assert pose1 == pose4 and rep1 == rep4
    for a, b in zip(r1, r4):
        assert np.array_equal(a['transform'], b['transform']) and a['n_corr'] == b['n_corr']
        za = np.load(tmp_path / 'flight1' / f"0_{a['src_frame']}_{a['ref_frame']}.npz")
        zb = np.load(tmp_path / 'flight4' / f"0_{a['src_frame']}_{a['ref_frame']}.npz")
        assert set(za.files) == set(zb.files)
        for k in za.files:
            assert np.array_equal(za[k], zb[k]), k

    # round 5: without .npz outputs the Tester's pipeline runs lock-step groups by default (staged pairs drawn four at a time on each
    # stream): same records, pose lines and report
    d = tmp_path / 'grouped'
    t = infer.Tester(cfg, state, str(d), save_npz=False, ransac=False, pairs_in_flight=2)
    assert t.pipeline.lockstep == 4
    recs = t.run(dataset.PairStager(dataset.ArrayPairDataset(pairs), depth=8))
    assert t.pipeline.last_stats['lockstep_groups'] >= 1
    assert open(d / '00_pose').read() == pose1 and t.summary.lines() == rep1
    for a, b in zip(r1, recs):
        assert np.array_equal(a['transform'], b['transform']) and a['n_corr'] == b['n_corr']


def test_one_pair_in_flight_through_the_stager_is_not_throttled(setup):
    """Round 4: with two reader threads the stager's pinned-buffer fills went through torch's CPU copy_, whose OpenMP teams (one
    per calling thread, as wide as the host) spin against a container's CPU quota: 30-70 ms per pair instead of 4.  The fills
    are numpy copies now and the harness threads keep to one OpenMP thread; a generous bound catches the class of problem."""
    import time
    from rdmnet_amd import dataset, infer
    cfg, state, distinct = setup
    pairs = [distinct[i % 5] + (np.eye(4),) for i in range(40)]
    t = infer.Tester(cfg, state, None, save_npz=False, pairs_in_flight=1)
    t.run(dataset.PairStager(dataset.ArrayPairDataset(pairs[:8]), depth=2, workers=2))  # warm-up
    t0 = time.perf_counter()
    recs = t.run(dataset.PairStager(dataset.ArrayPairDataset(pairs), depth=2, workers=2))
    ms = (time.perf_counter() - t0) * 1e3 / len(pairs)
    assert len(recs) == 48 and ms < 15.0, ms


def test_an_engine_error_inside_the_pipeline_reaches_the_caller(setup):
    from rdmnet_amd import pipeline
    cfg, state, distinct = setup
    p = pipeline.PairPipeline(cfg, state, pairs_in_flight=2)
    good = (torch.from_numpy(distinct[0][0]).cuda(), torch.from_numpy(distinct[0][1]).cuda())
    bad = (good[0], torch.zeros((0, 3), device='cuda'))  # an empty cloud: rdm_engine_run refuses it
    with pytest.raises(RuntimeError, match='rdm_engine_run'):
        p.run_pairs([good, good, bad, good])
    assert len(p.run_pairs([good, good])) == 2  # the pipeline is usable afterwards
