"""Event-table input for the MI355X path.

`read_events` returns what `run.py` of the reference builds with `pd.read_csv(sep='\\t', usecols=[session, item, time],
dtype={session: 'int32', item: 'str'})` (run.py:45-78) -- except that the item column is a pandas Categorical whose
categories are the item-id strings in order of first appearance and whose codes are therefore already the `ItemIdx` of
gru4rec.py:534-538.  The parse is native (`g4r_events_load`, all cores, no Python string objects per event); `GRU4Rec.fit`
and `evaluation.evaluate_gpu` recognise the categorical column and skip the hash joins.  Files the native parser does not
cover fall back to pandas (plain `str` column), with identical downstream results.
"""
import numpy as np
import pandas as pd

from . import _native


def _header(path):
    with open(path, 'rt') as fh:
        return fh.readline().rstrip('\r\n').split('\t')


def read_events(path, session_key='SessionId', item_key='ItemId', time_key='Time', threads=0, engine='auto'):
    """engine: 'auto' (native, pandas if the file needs it), 'native' (raise instead of falling back) or 'pandas'."""
    wanted = [session_key, item_key, time_key]
    if engine != 'pandas':
        got = _native.load_events(path, session_key, item_key, time_key, threads)
        if got is not None:
            header = _header(path)
            cols = {session_key: got['session'], time_key: got['time'],
                    item_key: pd.Categorical.from_codes(got['item_idx'], categories=pd.Index(got['item_ids'], dtype=object), validate=False)}
            return pd.DataFrame({c: cols[c] for c in sorted(wanted, key=header.index)}, copy=False)
        if engine == 'native':
            raise ValueError('{} needs the general CSV parser (quoted fields, missing values or non-int32 session ids)'.format(path))
    return pd.read_csv(path, sep='\t', usecols=wanted, dtype={session_key: 'int32', item_key: 'str'})


def is_categorical(col):
    return isinstance(col.dtype, pd.CategoricalDtype)


def first_appearance_index(col):
    """For a categorical item column: (item ids in order of first appearance, ItemIdx per row) -- what
    `ids = col.unique(); pd.Series(arange(len(ids)), index=ids)[col.values].values` gives for a str column."""
    codes = np.asarray(col.cat.codes.values)
    if (codes < 0).any():
        raise ValueError('missing item ids in the event table')
    cats = np.asarray(col.cat.categories.values, dtype=object)
    seen = pd.unique(codes)
    if len(seen) == len(cats) and (seen == np.arange(len(cats))).all():
        return cats, codes.astype(np.int64)
    remap = np.full(len(cats), -1, dtype=np.int64)
    remap[seen] = np.arange(len(seen))
    return cats[seen], remap[codes]


def lookup_item_index(col, itemidmap):
    """ItemIdx per row of a categorical item column under a trained model's itemidmap, -1 for unknown items."""
    cat_idx = itemidmap.reindex(col.cat.categories).values.astype(np.float64)
    cat_idx = np.where(np.isnan(cat_idx), -1, cat_idx).astype(np.int64)
    codes = np.asarray(col.cat.codes.values).astype(np.int64)
    return np.where(codes >= 0, cat_idx[np.maximum(codes, 0)], -1)
