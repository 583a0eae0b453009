"""Host-side pieces added in round 2 (no GPU): plan padding, width padding of the device layout, the reference's host sampler,
the bench launcher's rank environment, per-kernel byte accounting of the gather/scatter roofline object."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bench
from gru4rec_amd import _native
from gru4rec_amd.gru4rec import GRU4Rec, _pad4, _pad_cols, _strip_cols
from gru4rec_amd.plan import build_rank_plan, pad_plan

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sessions(n=60, n_items=50, seed=3):
    rng = np.random.RandomState(seed)
    lens = rng.randint(2, 8, size=n)
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum(lens)
    return off, rng.permutation(n), rng.randint(0, n_items, size=off[-1]).astype(np.int32)


def test_pad_plan_appends_noop_steps_only():
    off, order, items = _sessions()
    plan = build_rank_plan(off, order, items, 8, 16)
    padded = pad_plan(plan, plan['T'] + 7)
    assert padded['T'] == plan['T'] + 7 and len(padded['M']) == padded['T']
    assert (padded['M'][plan['T']:] == 0).all() and (padded['reset'][plan['T']:] == 0).all()
    for k in ('in_idx', 'out_idx', 'reset', 'M'):
        np.testing.assert_array_equal(padded[k][:plan['T']], plan[k])
    assert pad_plan(plan, plan['T']) is plan
    with pytest.raises(ValueError):
        pad_plan(plan, plan['T'] - 1)


@pytest.mark.parametrize('world', [2, 3, 4])
def test_padded_rank_plans_keep_every_event(world):
    """Sum over ranks of the events of the padded plans == events of all sessions (nothing is truncated away)."""
    off, order, items = _sessions(n=90)
    plans = [build_rank_plan(off, order, items, 8, 16, r, world) for r in range(world)]
    T = max(p['T'] for p in plans)
    plans = [pad_plan(p, T) for p in plans]
    assert all(p['T'] == T for p in plans)
    assert sum(int(p['M'].sum()) for p in plans) == int((np.diff(off) - 1).sum())


def test_width_padding_round_trip():
    assert [_pad4(x) for x in (1, 4, 5, 100, 101)] == [4, 4, 8, 100, 104]
    a = np.arange(3 * 2 * 5, dtype=np.float32).reshape(3, 10)        # 2 blocks of 5
    p = _pad_cols(a, 2, 5, 8)
    assert p.shape == (3, 16) and (p[:, 5:8] == 0).all() and (p[:, 13:] == 0).all()
    np.testing.assert_array_equal(_strip_cols(p, 2, 5, 8), a)
    g = GRU4Rec(layers=[10, 6], constrained_embedding=True)
    g.n_items = 7
    assert g._dev_spec('Wx', 0) == (6, 8, 3, 10, 12)          # constrained: input rows = top layer width
    assert g._dev_spec('Wx', 1) == (10, 12, 3, 6, 8)
    assert g._dev_spec('acc_Wrz', 0) == (10, 12, 2, 10, 12)
    assert g._dev_spec('vel_Wy', 0) == (7, 7, 1, 6, 8)
    h = GRU4Rec(layers=[10], constrained_embedding=False, embedding=0)
    h.n_items = 7
    assert h._dev_spec('Wx', 0) == (7, 7, 3, 10, 12)          # one-hot: Wx[0] is the item-row table


def test_cpu_sampler_is_the_references_call_sequence():
    """store_type='cpu': np.searchsorted(pop, np.random.rand(n_sample * length)) on the float64 cumulative table, or
    np.random.choice when sample_alpha == 0 (gru4rec.py:507-514) -- same calls, same global stream."""
    g = GRU4Rec(n_sample=6, sample_alpha=0.5)
    g.n_items = 9
    pop = np.arange(1, 10, dtype=np.float64) ** 0.5
    g._pop64 = pop.cumsum() / pop.sum()
    g._pop64[-1] = 1
    np.random.seed(5)
    got = g._cpu_samples(4)
    np.random.seed(5)
    want = np.searchsorted(g._pop64, np.random.rand(24)).reshape(4, 6)
    np.testing.assert_array_equal(got, want)
    g.sample_alpha = 0.0
    np.random.seed(7)
    got = g._cpu_samples(3)
    np.random.seed(7)
    np.testing.assert_array_equal(got, np.random.choice(9, size=18).reshape(3, 6))


def test_bench_refuses_mismatched_world_size_and_prices_sparse_bytes():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '4', '--steps', '1'], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and 'WORLD_SIZE (2) != --gpus (4)' in (r.stderr + r.stdout)
    cfg = bench.CONFIGS['cfg4']
    B, ns, D = cfg['batch_size'], cfg['n_sample'], cfg['layers'][-1]
    R, N = 2 * B + ns, B + ns
    assert bench.algorithmic_cost(cfg)['k_sparse_update']['bytes'] == (5 * R * D + 5 * N + R) * 4


def test_unknown_rank_mode_and_new_symbols_are_exported():
    assert _native.RANK_MODES['tiebreaking'] == 3
    lib = _native.lib()
    for name in ('g4r_sync_export', 'g4r_sync_import', 'g4r_comm_max_i64', 'g4r_comm_nranks', 'g4r_bench_rows', 'g4r_set_step_counters'):
        assert hasattr(lib, name)
