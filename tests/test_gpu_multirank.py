"""The multi-rank machinery that can be exercised on ONE MI355X: M = 0 padding steps, the staged dense path with the RCCL
all-reduce captured into the step graph (one-rank communicator), and the reconciliation of the GPU-local item tables with two
model handles standing in for two ranks (g4r_sync_export / g4r_sync_import: the kernels g4r_comm_sync_sparse runs around its
all-gather)."""
import numpy as np
import pytest

from gru4rec_amd import _native
from gru4rec_amd.plan import pad_plan

from test_gpu_parity import CASES, make_pair, random_plan

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('staged', [0, 1])
def test_padding_steps_touch_nothing(monkeypatch, staged):
    """A plan padded with M = 0 steps (ranks with fewer sessions) ends with exactly the parameters of the unpadded plan, and the
    padding steps cost 0."""
    if staged:
        monkeypatch.setenv('G4R_FORCE_STAGED', '1')
    kw = dict(CASES['bprmax_elu'])
    I, B, ns, T = 80, 12, 24, 40
    outs = []
    for pad in (0, 5):
        _, m = make_pair(I, B, ns, store_rows=200, use_graph=1, **dict(kw))
        if staged:
            m.comm_init(_native.comm_unique_id(), 1, 0)
        plan = pad_plan(random_plan(I, B, T, seed=13), T + pad)
        m.set_plan(plan)
        m.train_steps(0, T + pad)
        losses = m.get_losses(0, T + pad)
        assert (losses[T:] == 0).all()
        outs.append((losses[:T], m.get_param('Wy', (I, 12)), m.get_param('By', (I,)), m.get_param('acc_Wy', (I, 12)),
                     m.get_param('Wx', (12, 36), 0), m.get_param('Wh', (12, 12), 0), m.get_param('acc_Wh', (12, 12), 0)))
        m.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_staged_path_replays_from_a_graph_that_holds_the_allreduce(monkeypatch):
    """N > 1 data path with a one-rank communicator: gradient staging -> RCCL all-reduce -> k_dense_apply, 16 whole steps per
    graph replay (the all-reduce is captured), against the fused single-GPU step (on the merged k_update, whose dense tiles add the batch
    in the order of the staged gradients; k_update_l's order differs in the last bits: tests/test_gpu_parity.py compares that one)."""
    monkeypatch.setenv('G4R_LEAN_UPDATE', '0')
    kw = CASES['bprmax_mom_drop']
    I, B, ns, T = 80, 12, 24, 70
    plan = random_plan(I, B, T, seed=17)
    outs = []
    for staged in (0, 1):
        if staged:
            monkeypatch.setenv('G4R_FORCE_STAGED', '1')
        _, m = make_pair(I, B, ns, store_rows=200, use_graph=1, **dict(kw))
        if staged:
            m.comm_init(_native.comm_unique_id(), 1, 0)
            assert m.comm_nranks() == 1
        m.set_plan(plan)
        m.train_steps(0, T)
        if staged:
            assert m.get_debug('graph_mode', (1,))[0] == 1.0, 'RCCL was not captured into the step graph'
            m.comm_sync_sparse()
        outs.append((m.get_losses(0, T), m.get_param('Wy', (I, 16)), m.get_param('Wx', (16, 48), 0),
                     m.get_param('Bh', (48,), 0), m.get_param('acc_Wh', (16, 16), 0)))
        m.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('rule', [('sum', 'sum'), ('mean', 'sum'), None], ids=['sum_sum', 'mean_sum', 'default'])
@pytest.mark.parametrize('name', ['bprmax_mom_drop', 'xe_sep_embed'])
def test_item_table_reconciliation_with_two_handles(name, rule):
    """Two handles = two ranks.  After training on different shards, exporting both parts and importing [part0, part1] on both:
    replicas are bit-identical, every row equals base + delta0 / c + delta1 / c (fp32, in that order; c = 1 under the SUM rule, the
    number of ranks that touched the row under MEAN), a row only one rank trained keeps that rank's value under either rule (up to
    the fp32 rounding of base + (value - base)), untouched rows keep their bits; a second round starts from the new base.  The
    default (None) is parameters / velocities MEAN, optimizer statistics SUM (g4r_sync_set_rule)."""
    kw = dict(CASES[name])
    I, B, ns, T = 120, 12, 24, 30
    D = kw['layers'][-1]
    ms, bases = [], []
    for r in range(2):
        _, m = make_pair(I, B, ns, store_rows=200, use_graph=1, **dict(kw))      # identical initial weights on both
        m.sync_enable()
        if rule is not None:
            m.sync_set_rule(*rule)
        ms.append(m)
    prule, srule = rule if rule is not None else ('mean', 'sum')
    names = [('Wy', (I, D)), ('acc_Wy', (I, D)), ('By', (I,)), ('acc_By', (I,))]
    if kw.get('momentum', 0) > 0:
        names += [('vel_Wy', (I, D))]
    groups = [0]
    if not kw.get('constrained_embedding', False):
        names += [('E', (I, kw['embedding'])), ('acc_E', (I, kw['embedding']))]
        groups = [0, 1]
    for rnd in range(2):
        base = {n: ms[0].get_param(n, sh) for n, sh in names}
        for n, sh in names:
            np.testing.assert_array_equal(base[n], ms[1].get_param(n, sh))
        local = []
        for r, m in enumerate(ms):
            plan = random_plan(I // 2, B, T, seed=100 + 10 * rnd + r)      # items 0..59 only: the upper half is touched by samples alone
            if r == 1:
                plan['in_idx'] += I // 3
                plan['out_idx'] += I // 3
            m.set_plan(plan)
            m.train_steps(0, T)
            local.append({n: m.get_param(n, sh) for n, sh in names})
        parts_by_group = {}
        for g in groups:
            parts = [m.sync_export(g) for m in ms]
            parts_by_group[g] = parts
            assert all(len(p[0]) > 0 and (np.diff(p[0]) > 0).all() for p in parts)
            for m in ms:
                m.sync_import(parts, g)
        for n, sh in names:
            a, b = ms[0].get_param(n, sh), ms[1].get_param(n, sh)
            np.testing.assert_array_equal(a, b)
            d0, d1 = local[0][n] - base[n], local[1][n] - base[n]
            rows = lambda x: np.abs(x.reshape(I, -1)).max(axis=1)
            mean = (prule if not n.startswith('acc') else srule) == 'mean'
            if mean:      # rows both ranks rewrote (their ids are in both exported parts) take half of each delta
                pg = parts_by_group[1 if n.split('_')[-1] == 'E' else 0]
                both = np.zeros(I, dtype=bool)
                both[np.intersect1d(pg[0][0], pg[1][0])] = True
                shp = (I,) + (1,) * (base[n].ndim - 1)
                c = np.where(both.reshape(shp), np.float32(2), np.float32(1))
                np.testing.assert_array_equal(a, (base[n] + d0 / c) + d1 / c)
                assert both.any()
            else:
                np.testing.assert_array_equal(a, (base[n] + d0) + d1)
            only0 = (rows(d0) > 0) & (rows(d1) == 0)
            if n in ('Wy', 'E'):
                assert only0.any()
            np.testing.assert_allclose(a.reshape(I, -1)[only0], local[0][n].reshape(I, -1)[only0], rtol=1e-5, atol=2e-7)      # base + (value - base): error of the order of ulp(base)
        for g in groups:      # everything reconciled: nothing left to export
            assert all(len(m.sync_export(g)[0]) == 0 for m in ms)
    for m in ms:
        m.close()


@pytest.mark.parametrize('name', ['bprmax_mom_drop', 'xe_sep_embed'])
def test_dense_reconciliation_equals_the_part_exchange(name):
    """Small item tables are reconciled entirely on the device (g4r_comm_sync_sparse: pack [n_items][plane widths + 1] deltas -> one
    all-reduce -> apply; here the all-reduce is the in-process sum of g4r_virtual_sync_dense).  Against the packed-parts exchange of
    the same two ranks: the same values up to the rounding of base + (d0 + d1) / c vs (base + d0 / c) + d1 / c, replicas bit-identical,
    untouched rows keep their bits, the touched bytes are cleared (nothing left to export), a second round starts from the new base."""
    kw = dict(CASES[name])
    I, B, ns, T = 900, 12, 24, 30      # 1440 negative draws over 900 items: some rows stay untouched
    D = kw['layers'][-1]
    pairs = {}
    for way in ('parts', 'dense'):
        ms = []
        for r in range(2):
            _, m = make_pair(I, B, ns, store_rows=200, use_graph=1, **dict(kw))
            m.sync_enable()
            ms.append(m)
        pairs[way] = ms
    names = [('Wy', (I, D)), ('acc_Wy', (I, D)), ('By', (I,)), ('acc_By', (I,))]
    if kw.get('momentum', 0) > 0:
        names += [('vel_Wy', (I, D))]
    groups = [0]
    if not kw.get('constrained_embedding', False):
        names += [('E', (I, kw['embedding'])), ('acc_E', (I, kw['embedding']))]
        groups = [0, 1]
    for rnd in range(2):
        before = {n: pairs['dense'][0].get_param(n, sh) for n, sh in names}
        for way, ms in pairs.items():
            for r, m in enumerate(ms):
                plan = random_plan(I // 2, B, T, seed=100 + 10 * rnd + r)
                if r == 1:
                    plan['in_idx'] += I // 3
                    plan['out_idx'] += I // 3
                m.set_plan(plan)
                m.train_steps(0, T)
        touched = np.zeros(I, dtype=bool)
        for g in groups:
            parts = [m.sync_export(g) for m in pairs['parts']]
            if g == 0:
                for p in parts:
                    touched[p[0]] = True
            for m in pairs['parts']:
                m.sync_import(parts, g)
        _native.virtual_sync_dense(pairs['dense'])
        for n, sh in names:
            a, b = pairs['dense'][0].get_param(n, sh), pairs['dense'][1].get_param(n, sh)
            np.testing.assert_array_equal(a, b)
            want = pairs['parts'][0].get_param(n, sh)
            # round 0: one rounding of the combine; round 1 starts from bases that differ by that rounding and trains 30 more steps on it
            np.testing.assert_allclose(a, want, rtol=2e-6, atol=1e-7 if rnd == 0 else 2e-5)
            if n in ('Wy', 'acc_Wy', 'By', 'acc_By'):
                np.testing.assert_array_equal(a.reshape(I, -1)[~touched], before[n].reshape(I, -1)[~touched])
                assert touched.any() and (~touched).any()
        for g in groups:
            assert all(len(m.sync_export(g)[0]) == 0 for m in pairs['dense'])
    for ms in pairs.values():
        for m in ms:
            m.close()


def test_reconciliation_inside_train_steps_with_a_one_rank_communicator(monkeypatch):
    """g4r_set_sync_every: small item tables are reconciled by g4r_train_steps itself every k steps (pack -> all-reduce -> apply on the
    stream, between two steps, counted across calls).  One rank: the reconciliation is the identity up to the rounding of
    base + (value - base), so the run must track the one without it, the counter must show a reconciliation at every k-th step, and
    nothing may be left to export afterwards."""
    monkeypatch.setenv('G4R_FORCE_STAGED', '1')
    kw = CASES['bprmax_mom_drop']
    I, B, ns, T = 200, 12, 24, 40
    plan = random_plan(I, B, T, seed=23)
    outs = []
    for k in (0, 8):
        _, m = make_pair(I, B, ns, store_rows=200, use_graph=1, **dict(kw))
        m.comm_init(_native.comm_unique_id(), 1, 0)
        assert m.set_sync_every(k) == (k > 0)
        m.set_plan(plan)
        m.train_steps(0, 25)
        m.train_steps(25, T - 25)          # the count runs across calls: 8, 16, 24 | 32 (the one due at 40 waits for the next step)
        if k:
            assert int(m.get_debug('dev_syncs', (1,))[0]) == 4
            m.comm_sync_sparse()
            assert len(m.sync_export(0)[0]) == 0
        outs.append((m.get_losses(0, T), m.get_param('Wy', (I, 16)), m.get_param('acc_Wy', (I, 16)), m.get_param('Wx', (16, 48), 0)))
        m.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6)


def test_touched_rows_of_a_4_5_M_item_catalogue_come_back_as_the_sorted_id_list():
    """The touched bitmap is compacted ON THE DEVICE (k_touched_count / _scan / _write, g4r_sync_kernels.cuh): per 4096 items a
    workgroup count, an exclusive scan of the counts that runs in chunks of 1024 workgroups (4.5 M items = 1099 of them: the
    carry between chunks is exercised), ids written in item order.  Items at chunk and catalogue boundaries are planted among the
    inputs, targets and negatives; the exported list must be exactly the sorted set of rows the steps touched."""
    I, B, ns, T = 4_500_000, 8, 16, 6
    o, m = make_pair(I, B, ns, store_rows=T, layers=(4,), loss='bpr-max', final_act='linear', constrained_embedding=True, learning_rate=0.05)
    m.sync_enable()
    rng = np.random.RandomState(5)
    edge = np.array([0, 1, 4095, 4096, 4097, 8191, 1024 * 4096 - 1, 1024 * 4096, 1024 * 4096 + 1, 1025 * 4096 - 1, I - 4097, I - 2, I - 1], dtype=np.int64)
    plan = random_plan(I, B, T, seed=11)
    plan['in_idx'][0, :] = edge[:B]
    plan['out_idx'][1, :5] = edge[B:]
    ST = rng.randint(0, I, size=(T, ns)).astype(np.int32)
    ST[2, :edge.size] = edge
    m.set_sample_store(ST)
    m.set_plan(plan)
    m.train_steps(0, T)
    ids, rows = m.sync_export(0)
    want = np.unique(np.concatenate([plan['in_idx'][:T].ravel(), plan['out_idx'][:T].ravel(), ST[:T].ravel()]))
    np.testing.assert_array_equal(ids, want.astype(np.int32))
    assert rows.size == ids.size * int(_native.lib().g4r_sync_row_floats(m.h, 0))
    m.close()
