"""Recall@N / MRR@N on the device: same signature and results as the reference's
`evaluation.evaluate_gpu` (evaluation.py:15-147), without the Theano graph.

`evaluate_gpu` hands the whole test loop to the device in ONE call (`g4r_evaluate`): the session-parallel schedule
of evaluation.py:96-139 is the schedule of `fit` (same author, same loop), so the C++ plan builder produces it; every
step runs the GRU forward, scores all / the given items (final activation applied in fp32 as the reference does), ranks
the targets with the > / >= / == counts of :62-65 (mode 'tiebreaking': scores + uniform * 1e-10 in fp32 first, :55, from a Philox
stream keyed by the model seed, evaluation step, row and candidate column) and adds the hits and reciprocal ranks of every cut-off into device
accumulators; hidden rows of finished sessions are zeroed / dropped on the device.  `evaluate_gpu_stepwise` is the
host-driven variant (one `g4r_predict_step` + `g4r_rank_targets` per step), kept as a cross-check.
"""
import numpy as np
import pandas as pd


def _prepare_categorical(gru, test_data, items, session_key, item_key, time_key):
    """_prepare for tables from eventio.read_events: the inner join on the item id, the (session, time, item id) ordering and
    the session sizes of evaluation.py:86-95 as linear passes over integer arrays."""
    from . import datatools, eventio
    col = test_data[item_key]
    idx = eventio.lookup_item_index(col, gru.itemidmap)
    keep = idx >= 0                                   # how='inner': events of items the model does not know are dropped
    sess, tm = test_data[session_key].values[keep], test_data[time_key].values[keep]
    idx, codes = idx[keep], np.asarray(col.cat.codes.values)[keep]
    n = len(idx)
    ordered = n < 2 or bool(np.all((sess[1:] > sess[:-1]) | ((sess[1:] == sess[:-1]) & (tm[1:] > tm[:-1]))))
    if not ordered:
        # ties on (session, time) are broken by the item id *string*, as sort_values on a str column does
        cats = np.asarray(col.cat.categories.values, dtype=object).astype(str)
        lex = np.empty(len(cats), dtype=np.int64)
        lex[np.argsort(cats, kind='stable')] = np.arange(len(cats))
        order = np.lexsort((lex[codes], tm, sess))
        sess, idx = sess[order], idx[order]
    item_idxs = None if items is None else gru.itemidmap[items].values.astype(np.int32)
    offs = datatools.compute_offset(pd.DataFrame({session_key: sess}, copy=False), session_key).astype(np.int64)
    return idx.astype(np.int32), item_idxs, offs


def _prepare(gru, test_data, items, session_key, item_key, time_key):
    if isinstance(test_data[item_key].dtype, pd.CategoricalDtype):
        return _prepare_categorical(gru, test_data, items, session_key, item_key, time_key)
    lookup = pd.DataFrame({'ItemIdx': gru.itemidmap.values, item_key: gru.itemidmap.index})
    test_data = pd.merge(test_data, lookup, on=item_key, how='inner')
    test_data.sort_values([session_key, time_key, item_key], inplace=True)
    titems = test_data.ItemIdx.values.astype(np.int32)
    item_idxs = None if items is None else gru.itemidmap[items].values.astype(np.int32)
    sizes = test_data.groupby(session_key).size().values
    offs = np.zeros(len(sizes) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(sizes)
    return titems, item_idxs, offs


def evaluate_gpu(gru, test_data, items=None, session_key='SessionId', item_key='ItemId', time_key='Time',
                 cut_off=[20], batch_size=100, mode='standard'):
    """Returns (recall list, mrr list) -- one entry per cut-off, like the reference.  One device call for the whole test."""
    from . import _native
    if gru.error_during_train:
        raise Exception
    if mode not in ('standard', 'conservative', 'median', 'tiebreaking'):
        raise NotImplementedError
    multi = isinstance(cut_off, (list, tuple))
    cuts = list(cut_off) if multi else [cut_off]
    print('Measuring Recall@{} and MRR@{}'.format(','.join(str(c) for c in cuts), ','.join(str(c) for c in cuts)))
    model = gru._ensure_model()
    titems, item_idxs, offs = _prepare(gru, test_data, items, session_key, item_key, time_key)
    n_sessions = len(offs) - 1
    if n_sessions < batch_size:
        raise IndexError('fewer test sessions ({}) than batch_size ({})'.format(n_sessions, batch_size))
    # sessions in id order (evaluation.py:90-95); n_sample = 1 selects "run until no session is left" (:124-127)
    plan = _native.build_plan(offs.astype(np.int32), np.arange(n_sessions), titems, batch_size, 1)
    rec, mrr, n = model.evaluate(plan, batch_size, item_idxs, cuts, mode)
    # the evaluation used the model's prediction state; predict_next_batch starts afresh afterwards (the reference's
    # evaluate_gpu keeps an H of its own, evaluation.py:54)
    gru.predict = None
    return (rec / n).tolist(), (mrr / n).tolist()


def evaluate_gpu_stepwise(gru, test_data, items=None, session_key='SessionId', item_key='ItemId', time_key='Time',
                          cut_off=[20], batch_size=100, mode='standard'):
    """Host-driven variant of evaluate_gpu: the loop of evaluation.py:96-139 in Python, one device step at a time."""
    if gru.error_during_train:
        raise Exception
    if mode not in ('standard', 'conservative', 'median', 'tiebreaking'):
        raise NotImplementedError
    multi = isinstance(cut_off, (list, tuple))
    cuts = list(cut_off) if multi else [cut_off]
    print('Measuring Recall@{} and MRR@{}'.format(','.join(str(c) for c in cuts), ','.join(str(c) for c in cuts)))
    model = gru._ensure_model()
    lookup = pd.DataFrame({'ItemIdx': gru.itemidmap.values, item_key: gru.itemidmap.index})
    test_data = pd.merge(test_data, lookup, on=item_key, how='inner')
    test_data.sort_values([session_key, time_key, item_key], inplace=True)
    titems = test_data.ItemIdx.values.astype(np.int32)
    item_idxs = None if items is None else gru.itemidmap[items].values.astype(np.int32)
    sizes = test_data.groupby(session_key).size().values
    n_sessions = len(sizes)
    offs = np.zeros(n_sessions + 1, dtype=np.int64)
    offs[1:] = np.cumsum(sizes)
    if n_sessions < batch_size:
        raise IndexError('fewer test sessions ({}) than batch_size ({})'.format(n_sessions, batch_size))
    rec = np.zeros(len(cuts))
    mrr = np.zeros(len(cuts))
    n = 0
    model.predict_begin(batch_size)
    slot = np.arange(batch_size)
    next_free = batch_size - 1
    first = offs[slot].copy()
    last = offs[slot + 1].copy()
    while True:
        run = int((last - first).min())
        for i in range(run - 1):
            cur_in = titems[first + i]
            cur_out = titems[first + i + 1]
            m = len(slot)
            if item_idxs is None:
                model.predict_step(cur_in, None, want_scores=False)
                ranks = model.rank_targets(cur_out, 0, mode)
            else:
                model.predict_step(cur_in, np.concatenate([cur_out, item_idxs]), want_scores=False)
                ranks = model.rank_targets(np.arange(m, dtype=np.int32), m, mode)
            for j, c in enumerate(cuts):
                hit = ranks <= c
                rec[j] += hit.sum()
                mrr[j] += (hit / ranks).sum()
            n += m
        first = first + run - 1
        done = (last - first) <= 1
        n_done = int(done.sum())
        slot[done] = next_free + 1 + np.arange(n_done)
        next_free += n_done
        valid = slot < n_sessions
        if not valid.any():
            break
        refill = done & valid
        first[refill] = offs[slot[refill]]
        last[refill] = offs[slot[refill] + 1]
        # hidden rows of restarted slots are zeroed, rows of exhausted slots dropped (evaluation.py:134-139)
        keep = np.nonzero(valid)[0].astype(np.int32)
        model.predict_hidden(zero_mask=refill.astype(np.uint8) if len(refill) == batch_size else
                             np.pad(refill.astype(np.uint8), (0, batch_size - len(refill))),
                             keep_rows=keep if len(keep) < len(valid) else None)
        slot, first, last = slot[valid], first[valid], last[valid]
    gru.predict = None      # see evaluate_gpu
    rec = (rec / n).tolist()
    mrr = (mrr / n).tolist()
    return rec, mrr
