"""CPU checks of the kink handling of the exact-shape parity tests (tests/test_gpu_parity.py: kink_items / kink_twin / between):
the rows of items whose score lies on the jump of a piecewise final activation are BOUNDED by the oracle's two slopes, not dropped."""
import numpy as np

from oracle.model import OracleGRU4Rec
from test_gpu_parity import between, kink_items, kink_twin


def _oracle():
    o = OracleGRU4Rec(n_items=40, layers=(8,), batch_size=6, loss='bpr-max', final_act='elu-0.5', n_sample=10,
                      constrained_embedding=True, learning_rate=0.1, seed=3)
    rng = np.random.RandomState(0)
    o.set_popularity(rng.randint(1, 9, size=40))
    o.make_sample_store(10 * 4)
    o.H[0] = (rng.randn(6, 8) * 0.3).astype(np.float32)
    return o, rng


def test_a_score_of_exactly_zero_is_on_the_kink_and_the_twin_takes_the_other_slope():
    o, rng = _oracle()
    X = rng.randint(0, 40, size=6)
    Y = np.array([1, 2, 3, 4, 5, 6])
    o.Wy[7] = 0.0
    o.By[7] = 0.0                       # item 7 as a negative: s = h . 0 + 0 = exactly 0 in every row
    samples = np.array([7, 8, 9, 10, 11, 12, 13, 14, 15, 16])
    twin = kink_twin(o)
    _, dbg = o.train_step(X, Y, 6, np.zeros(6, dtype=np.uint8), samples=samples, return_debug=True)
    _, dbg2 = twin.train_step(X, Y, 6, np.zeros(6, dtype=np.uint8), samples=samples, return_debug=True)
    assert kink_items(o, dbg) == {7} == kink_items(twin, dbg2)
    # the two runs differ on that column's gradient by the ratio of the slopes (1 vs alpha = 0.5 at s = 0), nowhere else
    col = 6 + 0
    np.testing.assert_allclose(dbg2['ds'][:, col], 0.5 * dbg['ds'][:, col], rtol=1e-6)
    other = np.ones(dbg['ds'].shape[1], dtype=bool)
    other[col] = False
    np.testing.assert_array_equal(dbg2['ds'][:, other], dbg['ds'][:, other])
    assert abs(float(o.By[7] - twin.By[7])) > 0      # so the row the kink touches differs between the slopes


def test_scores_away_from_zero_are_not_kink_items():
    o, rng = _oracle()
    _, dbg = o.train_step(rng.randint(0, 40, size=6), rng.randint(0, 40, size=6), 6, np.zeros(6, dtype=np.uint8), return_debug=True)
    assert kink_items(o, dbg) == set()


def test_between_accepts_the_interval_and_rejects_a_wrong_row():
    a = np.array([[1.0, -2.0, 3.0]])
    b = np.array([[1.5, -1.0, 3.0]])
    errs = []
    between('inside', np.array([[1.2, -1.5, 3.0]]), a, b, 1e-3, 1e-4, errs)
    between('on the ends', a, a, b, 1e-3, 1e-4, errs)
    between('on the ends', b, a, b, 1e-3, 1e-4, errs)
    assert not errs
    between('outside', np.array([[1.2, -1.5, 3.1]]), a, b, 1e-3, 1e-4, errs)      # 3 % off where both slopes agree
    assert len(errs) == 1
    errs = []
    between('beyond the far slope', np.array([[1.6, -1.5, 3.0]]), a, b, 1e-3, 1e-4, errs)
    assert len(errs) == 1
