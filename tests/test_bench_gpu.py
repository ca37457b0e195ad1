"""GPU: bench.py's multi-rank path on ONE device.  `python bench.py --gpus 2` with no launcher self-spawns two ranks
(RDM_BENCH_SHARE_DEVICE=1 puts both on cuda:0, gloo carries the collectives); the JSON line must account for every
step of both ranks exactly once."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'MASTER_PORT')}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), *args], capture_output=True, text=True, timeout=timeout,
                       env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, p.stdout[-2000:]  # rank 0 prints ONE line
    return json.loads(lines[0])


def test_two_ranks_self_spawned_on_one_device():
    r = run_bench('--gpus', '2', '--steps', '8', '--warmup', '2', '--ramp-seconds', '0', '--streams', '2', '--pairs', '4',
                  '--host-steps', '4', '--full-steps', '4', '--api-steps', '4', '--no-cpu-baseline', '--dist-backend', 'gloo', env_extra={'RDM_BENCH_SHARE_DEVICE': '1'})
    assert r['n_gpus'] == 2 and r['steps'] == 8 and r['scaling'] == 'weak' and r['unit'] == 'pairs/s'
    # a step = one engine call = one lock-step group of config.pairs_per_step pairs (round 5; --lockstep 1: one pair)
    pps = r['config']['pairs_per_step']
    assert pps == r['config']['lockstep_pairs_per_stream'] >= 2 and r['config']['pairs_in_flight_per_gpu'] == 2 * pps
    assert r['records'] == {'gathered': 16 * pps, 'distinct_steps': 16 * pps, 'distinct_pairs': 4}
    assert r['registration']['pairs'] == 16 * pps
    assert r['value'] > 0 and abs(r['value'] - 16 * pps / (r['ms_per_step'] * 8 / 1e3)) < 1e-6 * r['value']
    assert r['one_pair_per_call']['value'] > 0
    assert r['host_to_host']['value'] > 0 and r['drop_in_api']['value'] > 0 and r['cpu_baseline'] is None
    # round 6: the contract-faithful side figures are measured in the headline's schedule, the one-pair-per-call ones beside them
    for key in ('host_to_host', 'full_tables', 'drop_in_api'):
        assert 'lock-step' in r[key]['schedule'] and r[key]['one_pair_per_call']['value'] > 0, key
    # every pair of the two ranks' streams came from a different step; with 4 distinct clouds and 2 ranks a rank's steps cycle
    # through all of them (the pair index does not depend on the world size alone)
    assert r['config']['lockstep_records_per_launch'] > 1.5


def test_single_rank_line_has_the_contract_fields():
    r = run_bench('--steps', '8', '--warmup', '2', '--ramp-seconds', '0', '--pairs', '2', '--host-steps', '4', '--full-steps', '4', '--api-steps', '8', '--no-cpu-baseline')
    assert r['config']['pairs_per_step'] >= 2 and r['config']['pairs_per_gpu'] == 8 * r['config']['pairs_per_step']
    assert r['roofline']['group_alone']['frac'] > 0 and r['roofline']['single_pair_alone']['frac'] > 0 and r['roofline']['launches'] > 0
    for key in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
                'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert key in r, key
    assert r['n_gpus'] == 1 and r['vs_baseline'] is None and r['dtype'] == 'f32' and 'workload' in r['config']
    rf = r['roofline']
    assert rf['bound'] in ('hbm', 'mfma') and rf['peak'] > 0 and abs(rf['frac'] - rf['achieved'] / rf['peak']) < 1e-9
    # round-stable side keys of the roofline, the all-13-tables figure beside `value`, and the product scheduler
    assert 'definition' in rf and rf['whole_layer']['group_alone']['frac'] > 0 and rf['by_form'] and rf['real_slots']['fill'] <= 1
    # round 6: no fraction of the HBM peak above 1 in the moved-bytes variants, whatever the form; a launch of
    # the timed region carries a FULL group's bytes
    def fracs(d, path=''):
        for k, v in (d.items() if isinstance(d, dict) else []):
            if k == 'moved_bytes' and isinstance(v, dict):
                yield path + '/' + k, v['frac']
            else:
                yield from fracs(v, path + '/' + k)
    moved = list(fracs(rf))
    assert moved and all(0 < f <= 1.0 for _, f in moved), moved
    assert abs(rf['bytes_per_launch'] / rf['group_alone']['bytes_per_launch'] - 1) < 0.25
    assert r['full_tables']['value'] > 0 and r['config']['searches_per_pair'] == 12
    assert r['config']['scheduler'] == 'rdmnet_amd.pipeline.PairPipeline'


def test_single_rank_through_rccl():
    """The RCCL path on the one GPU a test box has: world size 1, backend nccl; the barriers, the record all_gather and
    the all_reduce(MAX) of the elapsed time go through the process group (the multi-GPU run's collectives,
    geotransformer/engine/base_tester.py:70-76,123-128 in the reference)."""
    r = run_bench('--gpus', '1', '--force-dist', '--steps', '8', '--warmup', '2', '--ramp-seconds', '0', '--pairs', '2', '--host-steps', '4',
                  '--full-steps', '0', '--api-steps', '4', '--no-cpu-baseline', '--lockstep', '1', timeout=600)  # (one pair per call: rounds 1-4)
    assert r['config']['pairs_per_step'] == 1
    assert r['collective']['backend'] == 'nccl' and r['collective']['forced_single_rank'] and r['collective']['library'].startswith('RCCL')
    assert r['records'] == {'gathered': 8, 'distinct_steps': 8, 'distinct_pairs': 2}
    assert r['n_gpus'] == 1 and r['value'] > 0 and r['host_to_host']['value'] > 0
    assert r['collective']['preflight']['ok'] and r['config']['gpu_max_hw_queues'] == 8


def test_dry_run_builds_the_rccl_communicator_and_gathers_without_running_a_pair():
    """`bench.py --dry-run` over RCCL (world of one: RCCL refuses two ranks on one device): communicator, pre-flight gather and
    timing reduction, one line, no pair -- what an 8-GPU launch does before its first pair (tests/test_distributed.py runs the
    same flag with two gloo ranks).  A real run reports the same pre-flight under `collective`."""
    d = run_bench('--gpus', '1', '--force-dist', '--dry-run', timeout=300)
    assert d['dry_run'] and d['backend'] == 'nccl' and d['library'].startswith('RCCL') and d['preflight']['ok']
    assert d['gpu_max_hw_queues'] == 8 and d['host_cpus_per_rank'] > 0
