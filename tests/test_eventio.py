"""Input pipeline: the native TSV loader against pandas.read_csv as the reference uses it (run.py:45-78),
and the categorical fast paths of fit / evaluate_gpu against the reference's pandas expressions (gru4rec.py:534-541,585;
evaluation.py:86-95).  Host only."""
import os

import numpy as np
import pandas as pd
import pytest

from gru4rec_amd import _native, eventio, evaluation


def _pandas(path, sk='SessionId', ik='ItemId', tk='Time'):
    return pd.read_csv(path, sep='\t', usecols=[sk, ik, tk], dtype={sk: 'int32', ik: 'str'})


def _write(path, rows, header, eol='\n', trailing=True):
    body = eol.join('\t'.join(str(v) for v in r) for r in [header] + rows)
    with open(path, 'w', newline='') as fh:
        fh.write(body + (eol if trailing else ''))


def _table(rng, n_sessions, n_items, weird_ids=False):
    rows, t = [], 1000
    for s in rng.permutation(n_sessions)[: max(1, n_sessions)]:
        for _ in range(rng.randint(1, 9)):
            it = rng.randint(0, n_items)
            iid = ('it%d' % it if it % 3 else str(214000000 + it)) if not weird_ids else ['a b', 'ü%d' % it, '0%d' % it, '-7'][it % 4]
            t += rng.randint(0, 3)
            rows.append((int(s) + 1, iid, t))
    return rows


def _assert_same(native, ref, ik='ItemId'):
    assert list(native.columns) == list(ref.columns)
    assert len(native) == len(ref)
    for c in ref.columns:
        if c == ik:
            assert eventio.is_categorical(native[c])
            assert native[c].astype(object).tolist() == ref[c].tolist()
            ids, idx = eventio.first_appearance_index(native[c])
            want_ids = ref[c].unique()
            assert ids.tolist() == want_ids.tolist()                       # itemidmap order = order of first appearance
            want = pd.Series(np.arange(len(want_ids)), index=want_ids)[ref[c].values].values
            assert np.array_equal(idx, want)
        else:
            assert native[c].dtype == ref[c].dtype, c
            assert np.array_equal(native[c].values, ref[c].values), c


@pytest.mark.parametrize('threads', [1, -2, -5, -16])
@pytest.mark.parametrize('variant', ['plain', 'crlf', 'no_trailing_newline', 'extra_columns', 'reordered', 'blank_lines', 'odd_ids'])
def test_native_loader_equals_pandas(tmp_path, threads, variant):
    rng = np.random.RandomState(len(variant) * 7 + abs(threads))
    rows = _table(rng, 60, 25, weird_ids=(variant == 'odd_ids'))
    header = ['SessionId', 'ItemId', 'Time']
    path = str(tmp_path / 'events.tsv')
    if variant == 'extra_columns':
        rows = [(r[0], 'x', r[1], 3.5, r[2], 'tail') for r in rows]
        header = ['SessionId', 'junk', 'ItemId', 'price', 'Time', 'more']
    if variant == 'reordered':
        rows = [(r[2], r[1], r[0]) for r in rows]
        header = ['Time', 'ItemId', 'SessionId']
    if variant == 'blank_lines':
        rows = rows[:10] + [()] + rows[10:]
    _write(path, rows, header, eol='\r\n' if variant == 'crlf' else '\n', trailing=(variant != 'no_trailing_newline'))
    got = eventio.read_events(path, threads=threads, engine='native')
    _assert_same(got, _pandas(path))


def test_custom_column_names_and_float_time(tmp_path):
    path = str(tmp_path / 'e.tsv')
    _write(path, [(1, 'a', 10), (1, 'b', 11.5), (2, 'a', 12), (2, 'c', '1e3')], ['sid', 'iid', 'ts'])
    got = eventio.read_events(path, 'sid', 'iid', 'ts', engine='native')
    ref = _pandas(path, 'sid', 'iid', 'ts')
    assert got['ts'].dtype == np.float64 == ref['ts'].dtype
    _assert_same(got, ref, 'iid')


@pytest.mark.parametrize('rows', [
    [(1, '"quoted"', 5), (1, 'b', 6)],          # quoting rules
    [(1, '', 5), (1, 'b', 6)],                  # missing item id
    [(1, 'a', ''), (1, 'b', 6)],                # missing time
    [(3000000000, 'a', 5)],                     # session id beyond int32: pandas raises, the native parser declines
], ids=['quotes', 'empty_item', 'empty_time', 'session_overflow'])
def test_files_for_the_general_parser_are_declined(tmp_path, rows):
    path = str(tmp_path / 'e.tsv')
    _write(path, rows, ['SessionId', 'ItemId', 'Time'])
    assert _native.load_events(path, 'SessionId', 'ItemId', 'Time') is None
    with pytest.raises(ValueError):
        eventio.read_events(path, engine='native')


def test_auto_engine_falls_back_to_pandas(tmp_path):
    path = str(tmp_path / 'e.tsv')
    _write(path, [(1, '"q"', 5), (1, 'b', 6), (2, 'b', 7)], ['SessionId', 'ItemId', 'Time'])
    got = eventio.read_events(path)
    assert not eventio.is_categorical(got['ItemId'])
    assert got.equals(_pandas(path))


def test_errors(tmp_path):
    with pytest.raises(_native.NativeError, match='cannot open'):
        _native.load_events(str(tmp_path / 'nope.tsv'), 'SessionId', 'ItemId', 'Time')
    path = str(tmp_path / 'e.tsv')
    _write(path, [(1, 'a', 5)], ['SessionId', 'ItemId', 'Time'])
    with pytest.raises(_native.NativeError, match='column Stamp'):
        _native.load_events(path, 'SessionId', 'ItemId', 'Stamp')
    open(path, 'w').close()
    with pytest.raises(_native.NativeError, match='empty'):
        _native.load_events(path, 'SessionId', 'ItemId', 'Time')


def test_header_only_file(tmp_path):
    path = str(tmp_path / 'e.tsv')
    _write(path, [], ['SessionId', 'ItemId', 'Time'])
    got = _native.load_events(path, 'SessionId', 'ItemId', 'Time')
    assert len(got['session']) == 0 and got['item_ids'] == []


def test_first_appearance_index_after_filtering():
    """A categorical column whose rows were filtered / reordered: indices must follow the rows, not the stale categories."""
    col = pd.Series(pd.Categorical.from_codes([2, 0, 2, 3, 0], categories=['a', 'b', 'c', 'd']))
    ids, idx = eventio.first_appearance_index(col)
    assert ids.tolist() == ['c', 'a', 'd'] and idx.tolist() == [0, 1, 0, 2, 1]


class _FakeModel:
    def __init__(self, ids):
        self.itemidmap = pd.Series(np.arange(len(ids)), index=ids, name='ItemIdx')


@pytest.mark.parametrize('seed', range(6))
def test_evaluation_prepare_categorical_equals_merge_path(seed):
    """evaluation._prepare on a categorical table == the reference's merge / sort_values / groupby (evaluation.py:86-95),
    including unknown items (inner join), unsorted sessions and (session, time) ties broken by the item id string."""
    rng = np.random.RandomState(seed)
    ids = ['i%d' % i for i in rng.permutation(40)]
    gru = _FakeModel(ids[:30])                                  # ten item ids are unknown to the model
    n = 300
    sess = rng.randint(1, 25, size=n)
    tm = rng.randint(0, 12, size=n) if seed % 2 else np.arange(n)      # even seeds: no ties; odd: many ties
    codes = rng.randint(0, 40, size=n)
    if seed >= 4:                                               # already ordered input takes the no-sort branch
        o = np.lexsort((tm, sess)); sess, tm, codes = sess[o], np.arange(n), codes[o]
    cat_order = rng.permutation(40)                             # category order unrelated to itemidmap order
    cats = [ids[i] for i in cat_order]
    inv = np.empty(40, dtype=np.int64); inv[cat_order] = np.arange(40)
    plain = pd.DataFrame({'SessionId': sess, 'ItemId': [ids[c] for c in codes], 'Time': tm})
    catdf = pd.DataFrame({'SessionId': sess, 'ItemId': pd.Categorical.from_codes(inv[codes], categories=cats), 'Time': tm})
    want = evaluation._prepare(gru, plain, None, 'SessionId', 'ItemId', 'Time')
    got = evaluation._prepare(gru, catdf, ids[:5], 'SessionId', 'ItemId', 'Time')
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
    assert got[1].tolist() == [0, 1, 2, 3, 4]


def test_prepare_host_products_equal_reference_expressions(tmp_path):
    """The array expressions prepare() uses (bincount support, first-row session start) against the reference's groupby()s."""
    rng = np.random.RandomState(3)
    rows = sorted(_table(rng, 200, 50), key=lambda r: (r[0], r[2]))
    path = str(tmp_path / 'e.tsv')
    _write(path, rows, ['SessionId', 'ItemId', 'Time'])
    from gru4rec_amd import datatools
    data = eventio.read_events(path, engine='native')
    ref = _pandas(path)
    ids, idx = eventio.first_appearance_index(data['ItemId'])
    itemidmap = pd.Series(np.arange(len(ids)), index=ids)
    support_ref = ref.groupby('ItemId').size()[itemidmap.index.values].values          # gru4rec.py:539-541
    assert np.array_equal(np.bincount(idx, minlength=len(ids)), support_ref)
    offs = datatools.compute_offset(data, 'SessionId')
    start_ref = ref.groupby('SessionId')['Time'].min().values                          # gru4rec.py:585
    assert np.array_equal(data['Time'].values[offs[:-1]], start_ref)


def test_non_utf8_item_ids_are_left_to_pandas(tmp_path):
    path = str(tmp_path / 'e.tsv')
    with open(path, 'wb') as fh:
        fh.write(b'SessionId\tItemId\tTime\n1\tcaf\xe9\t1\n1\tb\t2\n')
    assert _native.load_events(path, 'SessionId', 'ItemId', 'Time') is None
