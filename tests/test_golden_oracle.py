"""Pins the CPU oracle against golden vectors produced by running the REFERENCE's own source
(/root/reference/gru4rec.py, evaluation.py) on a Theano stand-in -- see oracle/make_golden.py.

Tolerances: the golden run is float32 under torch's summation order, the oracle float32 under NumPy's:
per-step cost rtol 2e-4 (atol 2e-6); parameters atol 5e-5 + rtol 1e-3; Recall/MRR exact to 1e-9 (they are
ratios of integer counts unless a rank flips, which these fixtures do not provoke).
"""
import ast
import glob
import os

import numpy as np
import pandas as pd
import pytest

from oracle.driver import OraclePredictor, oracle_evaluate, oracle_fit

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', '*.npz')))


def load(path):
    g = np.load(path, allow_pickle=False)
    params = ast.literal_eval(str(g['params']))
    train = pd.DataFrame({c: g['train_' + c] for c in ('SessionId', 'ItemId', 'Time')})
    test = pd.DataFrame({c: g['test_' + c] for c in ('SessionId', 'ItemId', 'Time')})
    return g, params, train, test


def oracle_params(params):
    p = dict(params)
    p['layers'] = tuple(p['layers'])
    return p


@pytest.mark.parametrize('path', GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_reference_run(path):
    g, params, train, test = load(path)
    store_type = str(g['store_type']) if 'store_type' in g else 'gpu'
    run = oracle_fit(train, oracle_params(params), int(g['sample_store']), seed=int(g['seed']), store_type=store_type)
    m = run.model
    assert m.n_items == int(g['n_items'])
    assert list(run.itemidmap.index) == list(g['itemids'])
    assert len(run.costs) == len(g['costs'])
    np.testing.assert_allclose(run.costs, g['costs'], rtol=2e-4, atol=2e-6)
    for i in range(len(m.layers)):
        np.testing.assert_allclose(m.Wx[i], g['Wx%d' % i], rtol=1e-3, atol=5e-5)
        np.testing.assert_allclose(m.Wh[i], g['Wh%d' % i], rtol=1e-3, atol=5e-5)
        np.testing.assert_allclose(m.Wrz[i], g['Wrz%d' % i], rtol=1e-3, atol=5e-5)
        np.testing.assert_allclose(m.Bh[i], g['Bh%d' % i], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(m.Wy, g['Wy'], rtol=1e-3, atol=5e-5)
    np.testing.assert_allclose(m.By, g['By'].reshape(-1), rtol=1e-3, atol=5e-5)
    if 'E' in g:
        np.testing.assert_allclose(m.E, g['E'], rtol=1e-3, atol=5e-5)
    # prediction with hidden-state carry-over and a session change (gru4rec.py:712-717)
    pr = OraclePredictor(m, run.itemidmap, 4)
    p1 = pr.predict_next_batch(np.array([1, 2, 3, 4]), g['pred_in1'])
    p2 = pr.predict_next_batch(np.array([1, 2, 9, 4]), g['pred_in2'])
    np.testing.assert_allclose(p1, g['pred1'], rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(p2, g['pred2'], rtol=2e-3, atol=2e-5)
    for mode in ('standard', 'conservative', 'median'):
        rec, mrr = oracle_evaluate(m, run.itemidmap, test, cut_off=[1, 5, 20], batch_size=5, mode=mode)
        np.testing.assert_allclose(rec, g['recall_' + mode], atol=1e-9)
        np.testing.assert_allclose(mrr, g['mrr_' + mode], atol=2e-7)      # the reference sums float32 reciprocal ranks


def test_goldens_exist():
    assert len(GOLDEN) >= 4
