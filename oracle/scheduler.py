"""Session-parallel mini-batch scheduler, CPU restatement (TEST INFRASTRUCTURE).

Restates the host loop of the reference's `GRU4Rec.fit` (gru4rec.py:587-651) and of
`evaluation.evaluate_gpu` (evaluation.py:90-139) as generators that emit exactly the
per-step tuples the reference hands to its compiled Theano function:

    fit  : (in_idx[M], out_idx[M], M, reset[M])            gru4rec.py:603-623
    eval : (in_idx[M], out_idx[M], M)                      evaluation.py:103-110

plus the hidden-state maintenance events between steps (row compaction by
`valid_mask` gru4rec.py:647-651; zero + compaction evaluation.py:134-139).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import this.
"""
import numpy as np


def fit_schedule(offset_sessions, session_idx_arr, data_items, batch_size, n_sample):
    """Yield ('step', in_idx, out_idx, M, reset) and ('compact', valid_mask) events.

    Follows gru4rec.py:594-651.  Slots that finish get the next session in `session_idx_arr`
    order; when sessions run out the batch shrinks ('compact' carries the boolean row mask the
    reference applies to every H[i], gru4rec.py:647-651).  Terminates when no valid slot is left
    or fewer than two remain without additional negative samples (gru4rec.py:637).
    """
    n_sessions = len(offset_sessions) - 1
    B = batch_size
    slot_sess = np.arange(B)                       # `iters`
    next_free = slot_sess.max()                    # `maxiter`
    first = offset_sessions[session_idx_arr[slot_sess]].astype(np.int64)
    last = offset_sessions[session_idx_arr[slot_sess] + 1].astype(np.int64)
    while True:
        run = int((last - first).min())
        cur_out = data_items[first]
        for i in range(run - 1):
            cur_in = cur_out
            cur_out = data_items[first + i + 1]
            reset = (first + i + 1 == last - 1)
            yield ('step', cur_in, cur_out, len(slot_sess), reset)
        first = first + run - 1
        done = (last - first <= 1)
        n_done = int(done.sum())
        slot_sess[done] = next_free + np.arange(1, n_done + 1)
        next_free += n_done
        valid = (slot_sess < n_sessions)
        n_valid = int(valid.sum())
        if n_valid == 0 or (n_valid < 2 and n_sample == 0):
            return
        refill = done & valid
        sess = session_idx_arr[slot_sess[refill]]
        first[refill] = offset_sessions[sess]
        last[refill] = offset_sessions[sess + 1]
        slot_sess = slot_sess[valid]
        first = first[valid]
        last = last[valid]
        if n_valid < len(valid):
            yield ('compact', valid)


def eval_schedule(offset_sessions, test_items, batch_size):
    """Yield ('step', in_idx, out_idx, M) and ('hidden', zero_mask, valid_mask) events.

    Follows evaluation.py:90-139: sessions are taken in id order (no time sort); after each
    block the rows of finished-and-refilled slots are zeroed (`tmp[mask] = 0`, :137) and the rows
    of exhausted slots dropped (`tmp[valid_mask]`, :138).
    """
    n_sessions = len(offset_sessions) - 1
    slot_sess = np.arange(batch_size)
    next_free = slot_sess.max()
    first = offset_sessions[slot_sess].astype(np.int64)
    last = offset_sessions[slot_sess + 1].astype(np.int64)
    while True:
        run = int((last - first).min())
        cur_out = test_items[first]
        for i in range(run - 1):
            cur_in = cur_out
            cur_out = test_items[first + i + 1]
            yield ('step', cur_in, cur_out, len(slot_sess))
        first = first + run - 1
        done = (last - first <= 1)
        n_done = int(done.sum())
        slot_sess[done] = next_free + np.arange(1, n_done + 1)
        next_free += n_done
        valid = (slot_sess < n_sessions)
        if int(valid.sum()) == 0:
            return
        refill = done & valid
        sess = slot_sess[refill]
        first[refill] = offset_sessions[sess]
        last[refill] = offset_sessions[sess + 1]
        slot_sess = slot_sess[valid]
        first = first[valid]
        last = last[valid]
        yield ('hidden', refill, valid)
