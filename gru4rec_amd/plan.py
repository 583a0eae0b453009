"""Epoch-plan helpers on the host: session sharding for data-parallel runs (one process per GPU).

The reference has no multi-GPU mode; sessions are the natural unit (they are independent, and the reference already
advances `batch_size` of them in lock-step, gru4rec.py:594-651).  Rank r takes every nranks-th session of the
(time-sorted) session order, so every rank sees the same temporal progression."""
import numpy as np

from . import _native


def shard_sessions(offsets, order, data_items, rank, nranks):
    """Return (sub_offsets int32[n+1], sub_items int32[...]) holding only the sessions order[rank::nranks],
    re-numbered 0..n-1 in that order (so the scheduler's session order is simply arange(n))."""
    offsets = np.asarray(offsets)
    mine = np.asarray(order)[rank::nranks]
    lens = (offsets[1:] - offsets[:-1])[mine]
    sub_off = np.zeros(len(mine) + 1, dtype=np.int32)
    sub_off[1:] = np.cumsum(lens)
    if len(mine):
        idx = np.concatenate([np.arange(offsets[s], offsets[s + 1]) for s in mine])
    else:
        idx = np.zeros(0, dtype=np.int64)
    return sub_off, np.asarray(data_items)[idx].astype(np.int32)


def build_rank_plan(offsets, order, data_items, batch_size, n_sample, rank=0, nranks=1):
    """The (X, Y, M, R) stream of one epoch for this rank (gru4rec.py:594-651 over the rank's sessions)."""
    if nranks <= 1:
        return _native.build_plan(offsets, order, data_items, batch_size, n_sample)
    sub_off, items = shard_sessions(offsets, order, data_items, rank, nranks)
    return _native.build_plan(sub_off, np.arange(len(sub_off) - 1), items, batch_size, n_sample)


def pad_plan(plan, T):
    """Append M = 0 steps up to T steps.  Ranks of a data-parallel run hold plans of different lengths (their session shards differ);
    every rank has to issue the same number of dense-gradient all-reduces, so the shorter plans end with no-op steps: no row is
    active, the rank contributes a zero gradient and still applies the reduced one.  No event is dropped (the alternative, cutting
    every plan to the shortest, would drop the tail sessions of the longer ones)."""
    n = int(T) - int(plan['T'])
    if n < 0:
        raise ValueError('cannot pad a plan of %d steps to %d' % (plan['T'], T))
    if n == 0:
        return plan
    B = plan['in_idx'].shape[1]
    out = dict(plan)
    out['in_idx'] = np.concatenate([plan['in_idx'], np.zeros((n, B), dtype=np.int32)])
    out['out_idx'] = np.concatenate([plan['out_idx'], np.zeros((n, B), dtype=np.int32)])
    out['reset'] = np.concatenate([plan['reset'], np.zeros((n, B), dtype=np.uint8)])
    out['M'] = np.concatenate([plan['M'], np.zeros(n, dtype=np.int32)])
    out['T'] = int(T)
    return out
