"""Host-side table helpers used by `GRU4Rec.fit` before the epoch plan is built.  Same entry points and console
messages as the reference's datatools.py (`sort_if_needed`, `compute_offset`); pandas / NumPy only."""
import time

import numpy as np


def _first_unordered_key(table, keys, first_key_any_order):
    """None when `table` is lexicographically non-decreasing in `keys` (the first key may instead merely be grouped),
    otherwise the key at which the order breaks."""
    n = len(table)
    if n < 2:
        return None
    tie = np.ones(n - 1, dtype=bool)          # rows whose preceding keys are all equal to the previous row's
    for depth, key in enumerate(keys):
        col = table[key].values
        step_up, same = col[1:] > col[:-1], col[1:] == col[:-1]
        if depth == 0 and first_key_any_order:
            # grouped but not necessarily ascending: as many value changes as distinct values minus one
            if int((~same).sum()) + 1 != table[key].nunique():
                return key
        elif np.any(tie & ~step_up & ~same):
            return key
        tie &= same
    return None


def sort_if_needed(data, columns, any_order_first_dim=False):
    """Sorts `data` in place by `columns` unless it already is in that order."""
    broken_at = _first_unordered_key(data, columns, any_order_first_dim)
    if broken_at is None:
        print('The dataframe is already sorted by {}'.format(', '.join(columns)))
        return
    print('The dataframe is not sorted by {}, sorting now'.format(broken_at))
    began = time.time()
    data.sort_values(columns, inplace=True)
    print('Data is sorted in {:.2f}'.format(time.time() - began))


def compute_offset(data, column):
    """Start offset of every run of equal `column` values in the (sorted) table, plus the total length: int32[n_groups + 1]."""
    values = data[column].values
    if len(values) == 0:
        return np.zeros(1, dtype=np.int32)
    starts = np.flatnonzero(np.concatenate(([True], values[1:] != values[:-1])))
    return np.concatenate((starts, [len(values)])).astype(np.int32)
