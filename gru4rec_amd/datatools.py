"""Host-side data helpers with the interface of the reference's datatools.py (sort_if_needed :12-34,
compute_offset :36-39).  Pure pandas/NumPy; feeds the epoch-plan builder."""
import time

import numpy as np


def sort_if_needed(data, columns, any_order_first_dim=False):
    """Sort `data` in place by `columns` unless it already is (same messages as the reference)."""
    ok = True
    prev_neq = None
    col = columns[0]
    for pos, col in enumerate(columns):
        v = data[col].values
        neq = v[1:] != v[:-1]
        if pos == 0:
            if any_order_first_dim:
                ok = ok and (data[col].nunique() == int(neq.sum()) + 1)
            else:
                ok = ok and bool(np.all(v[1:] >= v[:-1]))
        else:
            ok = ok and bool(np.all(prev_neq | (v[1:] >= v[:-1])))
        prev_neq = neq
        if not ok:
            break
    if ok:
        print('The dataframe is already sorted by {}'.format(', '.join(columns)))
        return
    print('The dataframe is not sorted by {}, sorting now'.format(col))
    t0 = time.time()
    data.sort_values(columns, inplace=True)
    print('Data is sorted in {:.2f}'.format(time.time() - t0))


def compute_offset(data, column):
    """int32[n_groups + 1] start offsets of the (sorted) groups of `column`."""
    sizes = data.groupby(column).size().values
    offset = np.zeros(len(sizes) + 1, dtype=np.int32)
    offset[1:] = np.cumsum(sizes)
    return offset
