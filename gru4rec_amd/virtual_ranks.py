"""Virtual ranks: the data-parallel training semantics of an N-GPU run, computed on ONE GPU.

North_star: "sessions shard naturally across the 8 GPUs ... RCCL all-reduce ... of the dense GRU weight gradients each step
(sparse embedding rows stay GPU-local)".  The reference (gru4rec.py:594-651) is single-GPU, so what that does to the model is
not defined by it and has to be measured: N `GRU4Rec` objects of one process are prepared as ranks 0..N-1 (same initial
weights, session shard order[r::N], sample stream seed + 7919 r -- exactly what `set_distributed` gives a real rank), their device
handles are stepped in lock-step by `g4r_virtual_train_steps` (per step: every handle's kernels up to the dense gradients, the N
gradient buffers summed in rank order = the all-reduce, every handle's dense apply (/ N) and GPU-local sparse update), and the
item tables are reconciled with the two halves of `g4r_comm_sync_sparse` (`g4r_sync_export` / `g4r_sync_import`) at the end of
the epoch or every `sync_every` steps.  The result is bit for bit what N processes on N GPUs compute (RCCL's ring may sum the
gradients in another order: a rounding difference), only slower.  tests/test_gpu_virtual_ranks.py and
tools/virtual_ranks_study.py compare Recall@20 / MRR@20 (evaluation.py:62-75) of N = 2 / 8 with the single-rank run."""
import numpy as np

from . import _native
from .gru4rec import GRU4Rec
from .plan import build_rank_plan, pad_plan


def reconcile(models, groups=(0,), dense=False):
    """g4r_comm_sync_sparse without RCCL: every handle exports the rows it rewrote since the last call, every handle imports all
    parts in rank order; the replicas are bit-identical afterwards.  dense=True: the all-device form the library uses for small item
    tables (g4r_virtual_sync_dense; returns 0 rows, it never counts them)."""
    if dense:
        _native.virtual_sync_dense(models)
        return 0
    rows = 0
    for g in groups:
        parts = [m.sync_export(g) for m in models]
        rows += sum(len(p[0]) for p in parts)
        for m in models:
            m.sync_import(parts, g)
    return rows


def fit_virtual_ranks(params, data, nranks, sample_store=10000000, sync_every='default', chunk=64, on_chunk=None, rule=None, replicate=False,
                       sparse_exact=False):
    """Train `params['n_epochs']` epochs as `nranks` virtual ranks.  Returns (the rank objects -- rank 0 holds the reconciled weights
    on the host, ready for evaluate_gpu / predict_next_batch --, stats dict).  sync_every: reconcile the item tables every that
    many steps ('default': GRU4Rec.sync_every, what fit() does; None: only at the end of each epoch).  rule: (parameter rule, statistic rule) of the reconciliation,
    'sum' / 'mean' each (None: the library's default, g4r_sync_set_rule).  replicate: every rank gets ALL sessions and rank 0's
    sample stream instead of its shard -- N identical ranks must then reproduce the single-rank run (a check of this machinery).
    sparse_exact: the exact-replica mode (GRU4Rec.sparse_exact): per-occurrence gradient rows exchanged every step, nothing to reconcile."""
    grus = []
    for r in range(nranks):
        g = GRU4Rec(**params)
        g.sparse_exact = sparse_exact      # False / True (= 'reduce') / 'reduce' / 'mean' / 'sum'
        g.set_distributed(r, nranks, None)
        if replicate and not sparse_exact:
            g.seed -= 7919 * r      # _create_model adds 7919 * rank (exact-replica mode: it does not -- one sample stream for all ranks)
        g.prepare(data.copy(), sample_store=sample_store)
        exact = bool(getattr(g, 'sparse_exact', False)) and nranks > 1
        if nranks > 1 and not exact:
            g._model.sync_enable()
            if rule is not None:
                g._model.sync_set_rule(*rule)
        grus.append(g)
    if sync_every == 'default':
        sync_every = grus[0].sync_steps(nranks)
    if nranks <= 1 or exact:
        sync_every = None      # nothing to reconcile (one rank; or sparse_exact: the replicas never diverge)
    models = [g._model for g in grus]
    groups = (0,) if grus[0].constrained_embedding or not grus[0].embedding else (0, 1)
    stats = dict(steps=[], events=[], loss=[], sync_rows=0, syncs=0)
    for epoch in range(grus[0].n_epochs):
        plans = [build_rank_plan(g._offsets, g._base_order, g._data_items, g.batch_size, g.n_sample, 0 if replicate else r, 1 if replicate else nranks)
                 for r, g in enumerate(grus)]
        T = max(p['T'] for p in plans)
        plans = [pad_plan(p, T) for p in plans]
        for g, p in zip(grus, plans):
            g._model.set_plan(p)
            g._model.reset_hidden()
        done = since = 0
        while done < T:
            n = min(chunk, T - done)
            if sync_every:
                n = max(1, min(n, sync_every - since))
            if nranks > 1:
                _native.virtual_train_steps(models, done, n)
            else:
                models[0].train_steps(done, n)
            done += n
            since += n
            if nranks > 1 and sync_every and since >= sync_every and done < T:
                stats['sync_rows'] += reconcile(models, groups)
                stats['syncs'] += 1
                since = 0
            if on_chunk is not None:
                on_chunk(epoch, done, T)
        costs = [m.get_losses(0, T) for m in models]
        stats.setdefault('step_costs', []).append(costs)
        if any(np.isnan(c).any() for c in costs):
            raise FloatingPointError('NaN cost in a virtual-rank epoch')
        if nranks > 1 and not exact:
            stats['sync_rows'] += reconcile(models, groups)
            stats['syncs'] += 1
        Ms = [p['M'][:T] for p in plans]
        ev = float(sum(M.sum() for M in Ms))
        stats['steps'].append(int(T))
        stats['events'].append(int(ev))
        stats['loss'].append(float(sum((c * M).sum() for c, M in zip(costs, Ms)) / max(ev, 1.0)))
    for g in grus:
        g._download_weights()
    return grus, stats
