"""The N > 1 path on CPU: two processes (gloo) run the PRODUCT's host code for a data-parallel epoch -- session sharding and plan
building (gru4rec_amd.plan.build_rank_plan: the C++ scheduler), the common plan length (max over ranks) and the M = 0 padding
steps (pad_plan) -- with the device calls stood in for by the oracle: dense GRU gradients all-reduced (averaged) every step,
item rows rank-local and reconciled every SYNC_EVERY steps and at the end of the epoch by the product's rule (DESIGN.md section 7,
g4r_sync_kernels.cuh): for the rows some rank rewrote since the last reconciliation, parameters and velocities end at
base + MEAN over the touching ranks of (value - base), Adagrad's accumulators at base + SUM.

Checks: (1) the shards partition the sessions and preserve time order, and no event is lost: the events of all rank plans add
up to the events of the single-rank plan; (2) dense parameters stay bit-identical across ranks; (3) item-table replicas diverge
between reconciliations and agree after each one, and a row only one rank trained keeps exactly that rank's values (under the
mean as under the sum); (4) the two-process run equals a single-process emulation of the same algorithm (so the collectives sit
where the product puts them)."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gru4rec_amd import _native
from gru4rec_amd.plan import build_rank_plan, pad_plan, shard_sessions
from oracle.model import OracleGRU4Rec

I, B, NS, D = 60, 4, 8, 8
SYNC_EVERY = 4      # GRU4Rec.sync_every (16 in the product; the epoch here is ~20 steps)
PARAMS = dict(layers=(D,), batch_size=B, loss='bpr-max', final_act='elu-0.5', n_sample=NS, constrained_embedding=True,
              learning_rate=0.1, momentum=0.1)


def make_sessions(seed=5, n=40):
    rng = np.random.RandomState(seed)
    lens = rng.randint(2, 7, size=n)
    off = np.zeros(n + 1, dtype=np.int32)
    off[1:] = np.cumsum(lens)
    items = rng.randint(0, I, size=off[-1]).astype(np.int32)
    order = rng.permutation(n)
    return off, order, items


def make_rank_model(rank):
    o = OracleGRU4Rec(n_items=I, dtype=np.float64, seed=100 + 7919 * rank, **PARAMS)
    o.set_popularity(np.arange(1, I + 1))
    o.make_sample_store(NS * 6)
    return o


def planes(o):
    """(array, is_mean_plane) of the item-table group Wy / By: parameters and velocities take the mean, accumulators the sum"""
    return [(o.Wy, True), (o.By, True), (o.vel['Wy'], True), (o.vel['By'], True), (o.acc['Wy'], False), (o.acc['By'], False)]


def reconcile(models, bases, allreduce):
    """The product's reconciliation on a list of local replicas (one per rank in a process group: a list of one + a collective;
    the emulation: all replicas + a plain sum).  bases: per replica the plane values at the last reconciliation (updated in place)."""
    touched = [np.zeros(I, dtype=bool) for _ in models]
    for o, base, t in zip(models, bases, touched):
        for (cur, _), b0 in zip(planes(o), base):
            t |= (cur != b0).reshape(I, -1).any(axis=1)
    count = allreduce([t.astype(np.float64) for t in touched])
    hit = count > 0
    n_planes = len(bases[0])
    for q in range(n_planes):
        deltas = []
        for o, base, t in zip(models, bases, touched):
            cur = planes(o)[q][0]
            d = cur - base[q]
            d[~t] = 0
            deltas.append(d)
        total = allreduce(deltas)
        mean = planes(models[0])[q][1]
        for o, base in zip(models, bases):
            cur = planes(o)[q][0]
            div = np.where(count > 1, count, 1.0) if mean else np.ones(I)
            new = base[q] + total / (div if total.ndim == 1 else div[:, None])
            cur[hit] = new[hit]
            base[q][...] = cur
    return touched


def run_rank_steps(o, plan, T, reduce_fn, sync_fn):
    o.dense_grad_hook = reduce_fn
    for t in range(T):
        o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t])
        if (t + 1) % SYNC_EVERY == 0 or t + 1 == T:
            sync_fn(t)


def flat_dense(o):
    return np.concatenate([a.ravel() for n in ('Wx', 'Wh', 'Wrz', 'Bh') for a in getattr(o, n)])


def worker(rank, world, store_path, out_path):
    dist.init_process_group('gloo', init_method='file://' + store_path, rank=rank, world_size=world)
    off, order, items = make_sessions()
    plan = build_rank_plan(off, order, items, B, NS, rank, world)
    tt = torch.tensor([plan['T']])
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)          # g4r_comm_max_i64
    T = int(tt[0])
    plan = pad_plan(plan, T)                           # what GRU4Rec._epoch_plan does

    def allreduce_avg(grads):
        out = []
        for (i, *gs) in grads:
            red = []
            for g in gs:
                t = torch.from_numpy(np.ascontiguousarray(g))
                dist.all_reduce(t)                       # ncclAllReduce(sum) of the dense gradient buffer
                red.append(t.numpy() / world)            # k_dense_apply scales by 1/nranks
            out.append((i, *red))
        return out

    def allreduce_one(arrays):                           # this process holds one replica: the sum over ranks is a collective
        t = torch.from_numpy(np.ascontiguousarray(arrays[0]))
        dist.all_reduce(t)
        return t.numpy()
    o = make_rank_model(rank)
    base = [cur.copy() for cur, _ in planes(o)]          # identical on every rank (same initialisation)
    log = dict(local=[], synced=[], touched=[])

    def sync(t):
        log['local'].append(o.Wy.copy())
        log['touched'].append(reconcile([o], [base], allreduce_one)[0])
        log['synced'].append(np.concatenate([cur.reshape(I, -1) for cur, _ in planes(o)], axis=1))
    run_rank_steps(o, plan, T, allreduce_avg, sync)
    np.savez(out_path % rank, dense=flat_dense(o), local=np.array(log['local']), synced=np.array(log['synced']), touched=np.array(log['touched']),
             T=T, events=int(plan['M'].sum()))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_ranks_gloo_match_single_process_emulation():
    world = 2
    with tempfile.TemporaryDirectory() as d:
        store = os.path.join(d, 'store')
        out = os.path.join(d, 'rank%d.npz')
        mp.start_processes(worker, args=(world, store, out), nprocs=world, join=True, start_method='fork')
        r = [np.load(out % k) for k in range(world)]
    np.testing.assert_array_equal(r[0]['dense'], r[1]['dense'])          # (2) dense replicas identical
    n_sync = len(r[0]['synced'])
    assert n_sync >= 3
    np.testing.assert_array_equal(r[0]['synced'], r[1]['synced'])        # (3) replicas agree after EVERY reconciliation ...
    assert np.abs(r[0]['local'] - r[1]['local']).max() > 1e-6            # ... and had diverged before it
    only0 = r[0]['touched'] & ~r[1]['touched']
    both = r[0]['touched'] & r[1]['touched']
    assert only0.any() and both.any()
    for k in range(n_sync):                                              # a row only rank 0 trained keeps rank 0's own value
        np.testing.assert_allclose(r[0]['synced'][k][only0[k], :D], r[0]['local'][k][only0[k]], rtol=1e-14, atol=0)      # (base + delta: one rounding)
    # (1) no event is lost: all rank plans together hold the events of the single-rank plan
    off, order, items = make_sessions()
    single = _native.build_plan(off, order, items, B, NS)
    assert int(r[0]['events']) + int(r[1]['events']) == int(single['M'].sum()) == int((np.diff(off) - 1).sum())
    # (4) single-process emulation of the same algorithm
    plans = [build_rank_plan(off, order, items, B, NS, k, world) for k in range(world)]
    T = max(p['T'] for p in plans)
    assert T == int(r[0]['T']) and min(p['T'] for p in plans) < T        # the padding is exercised
    plans = [pad_plan(p, T) for p in plans]
    models = [make_rank_model(k) for k in range(world)]
    bases = [[cur.copy() for cur, _ in planes(m)] for m in models]
    emu_local = []
    for t in range(T):
        grads = []
        for k in range(world):
            m = models[k]
            captured = {}

            def cap(g, captured=captured):
                captured['g'] = g
                return g
            snap = {n: [a.copy() for a in getattr(m, n)] for n in ('Wx', 'Wh', 'Wrz', 'Bh', 'H')}
            state = (m.Wy.copy(), m.By.copy(), {k2: (v.copy() if isinstance(v, np.ndarray) else [a.copy() for a in v]) for k2, v in m.acc.items()},
                     {k2: (v.copy() if isinstance(v, np.ndarray) else [a.copy() for a in v]) for k2, v in m.vel.items()}, m.global_step, m.n_refills,
                     None if m.ST is None else m.ST.copy())
            m.dense_grad_hook = cap
            m.train_step(plans[k]['in_idx'][t], plans[k]['out_idx'][t], int(plans[k]['M'][t]), plans[k]['reset'][t])
            grads.append(captured['g'])
            # roll back, then replay below with the averaged gradient
            for n in ('Wx', 'Wh', 'Wrz', 'Bh', 'H'):
                setattr(m, n, snap[n])
            m.Wy, m.By, m.acc, m.vel, m.global_step, m.n_refills, m.ST = state
        avg = [(grads[0][j][0],) + tuple((grads[0][j][q] + grads[1][j][q]) / world for q in range(1, 5)) for j in range(len(grads[0]))]
        for k in range(world):
            models[k].dense_grad_hook = lambda g, avg=avg: avg
            models[k].train_step(plans[k]['in_idx'][t], plans[k]['out_idx'][t], int(plans[k]['M'][t]), plans[k]['reset'][t])
        if (t + 1) % SYNC_EVERY == 0 or t + 1 == T:
            emu_local.append(models[1].Wy.copy())
            reconcile(models, bases, lambda arrays: np.sum(arrays, axis=0))
    np.testing.assert_allclose(flat_dense(models[0]), r[0]['dense'], rtol=0, atol=1e-12)
    for k in range(world):
        got = np.concatenate([cur.reshape(I, -1) for cur, _ in planes(models[k])], axis=1)
        np.testing.assert_allclose(got, r[k]['synced'][-1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.array(emu_local), r[1]['local'], rtol=0, atol=1e-12)


def test_shards_partition_sessions_in_time_order():
    off, order, items = make_sessions()
    seen = []
    for k in range(3):
        sub_off, sub_items = shard_sessions(off, order, items, k, 3)
        mine = order[k::3]
        assert len(sub_off) == len(mine) + 1
        for j, s in enumerate(mine):
            np.testing.assert_array_equal(sub_items[sub_off[j]:sub_off[j + 1]], items[off[s]:off[s + 1]])
        seen += list(mine)
    assert sorted(seen) == list(range(len(order)))
