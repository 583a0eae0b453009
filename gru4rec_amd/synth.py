"""Deterministic RSC15-*shaped* synthetic click-stream generator (no datasets ship with this repo).

Shape targets: I = 37,483 items, session length 2 + Geometric(p=0.34) clipped at
200 (mean ~3.9), Zipf-Mandelbrot item popularity p(rank) ~ (rank + 27)^-1 (the most popular item holds ~0.5 % of the
events, as in RSC15; a pure Zipf(1.0) head would give it 9 %), and a first-order Markov structure (every item has a handful
of preferred successors) so that Recall@20 is learnable and sensitive to bugs.  Session start times
increase with the session id; events inside a session are 1..60 s apart.
"""
import numpy as np
import pandas as pd


def make_sessions(n_sessions, n_items=37483, seed=42, p_len=0.34, max_len=200, n_succ=10, p_follow=0.75,
                  zipf_s=1.0, zipf_q=27.0):
    """Return a DataFrame with columns SessionId (int32), ItemId (int64), Time (int64), sorted by session, time."""
    rng = np.random.RandomState(seed)
    lens = np.minimum(2 + rng.geometric(p_len, size=n_sessions) - 1, max_len).astype(np.int64)
    n_events = int(lens.sum())
    ranks = np.arange(1, n_items + 1, dtype=np.float64)
    pop = (ranks + zipf_q) ** (-zipf_s)
    pop /= pop.sum()
    cum = np.cumsum(pop)
    cum[-1] = 1.0
    perm = rng.permutation(n_items)                       # popularity rank -> item index
    succ = perm[np.minimum(np.searchsorted(cum, rng.rand(n_items, n_succ)), n_items - 1)]
    offs = np.zeros(n_sessions + 1, dtype=np.int64)
    offs[1:] = np.cumsum(lens)
    items = np.empty(n_events, dtype=np.int64)
    cur = perm[np.minimum(np.searchsorted(cum, rng.rand(n_sessions)), n_items - 1)]
    items[offs[:-1]] = cur
    alive = np.arange(n_sessions)
    pos = 1
    while len(alive):
        alive = alive[lens[alive] > pos]
        if not len(alive):
            break
        follow = rng.rand(len(alive)) < p_follow
        nxt_markov = succ[cur[alive], rng.randint(0, n_succ, size=len(alive))]
        nxt_pop = perm[np.minimum(np.searchsorted(cum, rng.rand(len(alive))), n_items - 1)]
        nxt = np.where(follow, nxt_markov, nxt_pop)
        items[offs[alive] + pos] = nxt
        cur[alive] = nxt
        pos += 1
    sess = np.repeat(np.arange(n_sessions, dtype=np.int32), lens)
    start = np.cumsum(rng.randint(1, 20, size=n_sessions)).astype(np.int64) + 1_400_000_000
    within = np.arange(n_events, dtype=np.int64) - np.repeat(offs[:-1], lens)
    times = np.repeat(start, lens) + within * 30 + rng.randint(0, 30, size=n_events)
    # external item ids are not the dense indices (as in real data)
    item_ids = (items + 1) * 7 + 214_500_000
    return pd.DataFrame({'SessionId': sess, 'ItemId': item_ids, 'Time': times})


def train_test_split(data, test_frac=1.0 / 30, min_len=2):
    """Mirror of examples/rsc15/preprocess.py:27-35: the last sessions (by start time) become the test set,
    test items are restricted to train items, test sessions shorter than 2 are dropped."""
    tmax = data.groupby('SessionId').Time.max()
    cut = tmax.quantile(1.0 - test_frac)
    train_ids = tmax[tmax < cut].index
    test_ids = tmax[tmax >= cut].index
    train = data[np.isin(data.SessionId, train_ids)].copy()
    test = data[np.isin(data.SessionId, test_ids)]
    test = test[np.isin(test.ItemId, train.ItemId)]
    tl = test.groupby('SessionId').size()
    test = test[np.isin(test.SessionId, tl[tl >= min_len].index)].copy()
    return train, test
