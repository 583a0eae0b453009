"""Parity of the HIP hot path against the CPU oracle, through the C ABI (ctypes).  Needs an MI355X.

Floating point tolerance (fp32 path vs fp32 oracle; both accumulate in fp32 but in different orders):
  per-step cost            rtol 2e-4 (+ atol 2e-6)
  intermediates            atol 3e-5 + rtol 3e-4 unless stated
  parameters               compared as UPDATES (value - initial value): |err| <= 1e-3 |update| + 1e-4 max|update| of the tensor
  optimizer accumulators   |err| <= 2e-4 |acc| + 1e-5 max|acc| of the tensor (a bias accumulator is the square of a column sum of
                           512 signed terms: cancellation leaves 1e-4 relative on its smallest entries, measured at cfg4)
(bounds relative to the tensor's own scale: an accumulator of 1e-7 that is off by 1 %, or an update that is off by 1 % of a
step, fails -- tests/test_gpu_mutation.py builds such libraries and expects red; runs of dozens of steps widen the bounds by a
stated factor, `loosen`, because the training dynamics amplify rounding differences.)
Integer work (sample store, plan, occurrence lists) is bit-exact.
"""
import os

import numpy as np
import pytest

from gru4rec_amd import _native
from oracle.model import OracleGRU4Rec, parse_act
from oracle.scheduler import fit_schedule

pytestmark = pytest.mark.gpu

REPORT = os.environ.get('G4R_PARITY_REPORT') or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'parity_report.txt')


def report(line):
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, 'a') as f:
            f.write(line + '\n')
    except OSError:
        pass


def close(name, got, want, atol=3e-5, rtol=3e-4, errs=None):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    worst = float((err / tol).max()) if err.size else 0.0
    bad = not np.isfinite(got).all() or worst > 1.0
    report('%-28s max_abs_err %.3e  max|want| %.3e  worst/tol %.3f %s' % (
        name, float(err.max()) if err.size else 0.0, float(np.abs(want).max()) if want.size else 0.0, worst,
        'FAIL' if bad else 'ok'))
    if bad and errs is not None:
        errs.append(name)
    return not bad


def close_rel(name, got, want, rtol, atol_frac, errs=None, floor=0.0):
    """|got - want| <= rtol |want| + atol_frac max|want| (+ floor): a bound relative to the tensor's own scale."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    err = np.abs(got - want)
    scale = float(np.abs(want).max()) if want.size else 0.0
    tol = rtol * np.abs(want) + atol_frac * scale + floor
    worst = float((err / np.maximum(tol, 1e-300)).max()) if err.size else 0.0
    bad = not np.isfinite(got).all() or worst > 1.0
    report('%-28s max_abs_err %.3e  max|want| %.3e  worst/tol %.3f %s' % (
        name, float(err.max()) if err.size else 0.0, scale, worst, 'FAIL' if bad else 'ok'))
    if bad and errs is not None:
        errs.append(name)
    return not bad


def snapshot(o):
    """Copies of every trainable array of the oracle (the initial values the updates are measured from)."""
    d = {'Wy': o.Wy.copy(), 'By': o.By.copy()}
    if o.E is not None:
        d['E'] = o.E.copy()
    for n in ('Wx', 'Wh', 'Wrz', 'Bh'):
        d[n] = [a.copy() for a in getattr(o, n)]
    return d


def make_pair(I, B, ns, store_rows, seed=3, **kw):
    """An oracle and a device model with identical weights / popularity / sample store."""
    use_graph = kw.pop('use_graph', 0)
    rank, nranks = kw.pop('rank', 0), kw.pop('nranks', 1)      # a handle that is one rank of several (same weights, same sample stream)
    sparse_exact = kw.pop('sparse_exact', 0)
    o = OracleGRU4Rec(n_items=I, batch_size=B, n_sample=ns, dtype=np.float32, seed=seed, **kw)
    rng = np.random.RandomState(seed)
    support = rng.randint(1, 40, size=I)
    o.set_popularity(support)
    o.make_sample_store(store_rows * ns if ns else 0)
    fa = parse_act(kw.get('final_act', 'linear'))
    ha = parse_act(kw.get('hidden_act', 'tanh'))
    m = _native.Model(
        n_items=I, layers=list(o.layers), batch_size=B, n_sample=ns, loss=_native.LOSS_IDS[o.loss],
        final_act=_native.ACT_IDS[fa[0]], final_act_p0=fa[1], final_act_p1=fa[2],
        hidden_act=_native.ACT_IDS[ha[0]], hidden_act_p0=ha[1], hidden_act_p1=ha[2],
        embed_mode=0 if o.constrained_embedding else (1 if o.embedding else 2), embedding=int(o.embedding or 0),
        learning_rate=o.learning_rate, momentum=o.momentum, lmbd=o.lmbd, bpreg=o.bpreg, logq=o.logq,
        smoothing=float(o.smoothing), adapt=_native.ADAPT_IDS[o.adapt],
        adapt_p0=float(o.adapt_params[0]) if len(o.adapt_params) > 0 else 0.0,
        adapt_p1=float(o.adapt_params[1]) if len(o.adapt_params) > 1 else 0.0, grad_cap=float(o.grad_cap),
        sample_alpha=o.sample_alpha, dropout_p_hidden=o.dropout_p_hidden, dropout_p_embed=o.dropout_p_embed,
        sample_store=store_rows * ns if ns else 0, seed=seed, device=0, rank=rank, nranks=nranks,
        use_graph=use_graph, sparse_exact=sparse_exact)
    # non-trivial biases / hidden state so that every term is exercised
    for i, D in enumerate(o.layers):
        o.Bh[i] = (rng.randn(3 * D) * 0.1).astype(np.float32)
        o.H[i] = (rng.randn(B, D) * 0.3).astype(np.float32)
        m.set_param('Wx', o.Wx[i], i)
        m.set_param('Wh', o.Wh[i], i)
        m.set_param('Wrz', o.Wrz[i], i)
        m.set_param('Bh', o.Bh[i], i)
        m.set_param('H', o.H[i], i)
    o.By = (rng.randn(I) * 0.1).astype(np.float32)
    m.set_param('Wy', o.Wy)
    m.set_param('By', o.By)
    if o.E is not None:
        m.set_param('E', o.E)
    m.set_popularity(o.P, o.lq_tgt if o.logq else None, o.lq_smp if o.logq else None)
    o.init0 = snapshot(o)
    return o, m


def random_plan(I, B, T, seed, tail=False):
    rng = np.random.RandomState(seed)
    M = np.full(T, B, dtype=np.int32)
    if tail:
        M[T // 2:] = max(1, B - 3)
        M[-1] = max(1, B // 2)
    return dict(in_idx=rng.randint(0, I, size=(T, B)).astype(np.int32),
                out_idx=rng.randint(0, I, size=(T, B)).astype(np.int32),
                reset=(rng.rand(T, B) < 0.25).astype(np.uint8), M=M, T=T, n_compact=0,
                compact_steps=np.zeros(0, dtype=np.int64), compact_maps=np.zeros((0, B), dtype=np.int32))


PIECEWISE = ('relu', 'leaky', 'elu', 'selu')


def kink_items(o, dbg):
    """Items of the step's score columns that hold a score ON THE KINK of a piecewise final activation (oracle debug record of a
    step): |s| within a few ulps of the score's own terms (oracle/model.py: kink_ulps * eps32 * (sum |h w| + |b|)), i.e. a score
    whose sign depends on the summation order.  The activations' derivative JUMPS there (gru4rec.py:189-223: T.switch(X >= 0)): an
    fp32 score that is exactly 0 in one order and -1e-10 in another gets a gradient that differs by the factor alpha, and Adagrad
    turns that one element into an update difference of ~1e-4 of the tensor's scale for that item.  Neither side is wrong (round 5:
    the oracle's score was EXACTLY 0.0 at row 93, column 2080 of step 5 of the configs[1]-shape test, the GPU's +-1e-10 depending on
    the last bits of the parameters).  Those items' rows are NOT left out: compare_params bounds them by the oracle's two slopes
    (a twin oracle that takes the other branch on exactly those elements, kink_twin)."""
    if str(o.final_act[0]) not in PIECEWISE:
        return set()
    return set(int(i) for i in np.asarray(dbg['Yp'])[np.asarray(dbg['kink_cols'], dtype=np.int64)])


def kink_twin(o):
    """A copy of the (not yet stepped) oracle that takes the OTHER slope on every score on the kink; None when the final activation
    has no kink."""
    if str(o.final_act[0]) not in PIECEWISE:
        return None
    import copy
    t = copy.deepcopy(o)
    t.kink_flip = True
    return t


def oracle_steps(o, plan, T, full_batch=None, twin=None):
    """The oracle over steps 0 .. T-1 of the plan: (per-step costs, items on the final activation's kink in any of the steps).
    twin (kink_twin) runs the same steps on the other slope."""
    costs, kink = [], set()
    for t in range(T):
        M = int(plan['M'][t]) if full_batch is None else full_batch
        cost, dbg = o.train_step(plan['in_idx'][t], plan['out_idx'][t], M, plan['reset'][t], return_debug=True)
        costs.append(cost)
        if dbg is not None:
            kink |= kink_items(o, dbg)
        if twin is not None:
            _, dbg2 = twin.train_step(plan['in_idx'][t], plan['out_idx'][t], M, plan['reset'][t], return_debug=True)
            if dbg2 is not None:
                kink |= kink_items(twin, dbg2)
    return costs, kink


def between(name, got, a, b, rtol, atol_rel, errs, floor=0.0):
    """got must lie between the two references a and b (element-wise), with close_rel's tolerance outside the interval."""
    got, a, b = (np.asarray(x, dtype=np.float64) for x in (got, a, b))
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    scale = max(float(np.abs(a).max()) if a.size else 0.0, float(np.abs(b).max()) if b.size else 0.0)
    tol = rtol * np.maximum(np.abs(a), np.abs(b)) + max(atol_rel * scale, floor)
    excess = np.maximum(lo - got, 0.0) + np.maximum(got - hi, 0.0)
    worst = float((excess / np.maximum(tol, 1e-300)).max()) if excess.size else 0.0
    ok = bool(np.all(excess <= tol)) and bool(np.isfinite(got).all())
    report('%-28s outside the two slopes by %.3e  scale=%.3e  worst/tol=%.3f  %s' % (name, float(excess.max()) if excess.size else 0.0, scale, worst, 'ok' if ok else 'FAIL'))
    if not ok:
        errs.append('%s: outside [min, max] of the two slopes by %.3e (scale %.3e, worst / tolerance %.2f)' % (name, float(excess.max()), scale, worst))


def compare_params(o, m, errs, tag, Mrows=None, loosen=1.0, init=None, skip_items=(), twin=None):
    """Parameters as updates against their initial values (o.init0, taken by make_pair), accumulators / velocities against their
    own scale.  loosen widens every bound by that factor (runs of many steps).  skip_items (kink_items) + twin (kink_twin, stepped
    by oracle_steps): the rows of items with a score on the final activation's kink are compared apart -- every element must lie
    between the oracle's two slopes (the run on `o` and the run on `twin`), with the same tolerance outside that interval; a row
    that is wrong by more than the kink explains fails like any other."""
    I = o.n_items
    Mrows = o.batch_size if Mrows is None else Mrows
    init = init if init is not None else o.init0
    PR, PA, AR, AA = 1e-3 * loosen, 1e-4 * loosen, 2e-4 * loosen, 1e-5 * loosen

    def upd(name, got, want, w0):
        w0 = np.asarray(w0, dtype=np.float64)
        # floor: a few ulps of the parameter itself (the update is the difference of two fp32 values)
        floor = 4.0 * float(np.spacing(np.float32(max(np.abs(np.asarray(want)).max(), 1e-30))))
        close_rel(name, np.asarray(got, dtype=np.float64) - w0, np.asarray(want, dtype=np.float64) - w0, PR, PA, errs, floor)

    for i, D in enumerate(o.layers):
        n_in = o.Wx[i].shape[0]
        upd('%s dWx%d' % (tag, i), m.get_param('Wx', (n_in, 3 * D), i), o.Wx[i], init['Wx'][i])
        upd('%s dWh%d' % (tag, i), m.get_param('Wh', (D, D), i), o.Wh[i], init['Wh'][i])
        upd('%s dWrz%d' % (tag, i), m.get_param('Wrz', (D, 2 * D), i), o.Wrz[i], init['Wrz'][i])
        upd('%s dBh%d' % (tag, i), m.get_param('Bh', (3 * D,), i), o.Bh[i], init['Bh'][i])
        close_rel('%s H%d' % (tag, i), m.get_param('H', (o.batch_size, D), i)[:Mrows], o.H[i][:Mrows], 3e-4 * loosen, 1e-4 * loosen, errs)
        for n in ('Wx', 'Wh', 'Wrz', 'Bh'):
            shape = {'Wx': (n_in, 3 * D), 'Wh': (D, D), 'Wrz': (D, 2 * D), 'Bh': (3 * D,)}[n]
            close_rel('%s acc_%s%d' % (tag, n, i), m.get_param('acc_' + n, shape, i), o.acc[n][i], AR, AA, errs)
    keep = np.ones(I, dtype=bool)
    keep[list(skip_items)] = False
    upd('%s dWy' % tag, m.get_param('Wy', (I, o.layers[-1]))[keep], o.Wy[keep], init['Wy'][keep])
    upd('%s dBy' % tag, m.get_param('By', (I,))[keep], o.By[keep], init['By'][keep])
    close_rel('%s acc_Wy' % tag, m.get_param('acc_Wy', (I, o.layers[-1]))[keep], o.acc['Wy'][keep], AR, AA, errs)
    close_rel('%s acc_By' % tag, m.get_param('acc_By', (I,))[keep], o.acc['By'][keep], AR, AA, errs)
    rows = sorted(int(i) for i in skip_items)
    if rows:
        assert twin is not None, 'kink items need the twin oracle (kink_twin)'
        gWy, gBy = m.get_param('Wy', (I, o.layers[-1]))[rows], m.get_param('By', (I,))[rows]
        w0, b0 = np.asarray(init['Wy'][rows], dtype=np.float64), np.asarray(init['By'][rows], dtype=np.float64)
        fl = lambda want: 4.0 * float(np.spacing(np.float32(max(np.abs(np.asarray(want)).max(), 1e-30))))
        between('%s dWy (kink rows)' % tag, gWy - w0, o.Wy[rows] - w0, twin.Wy[rows] - w0, PR, PA, errs, fl(o.Wy[rows]))
        between('%s dBy (kink rows)' % tag, gBy - b0, o.By[rows] - b0, twin.By[rows] - b0, PR, PA, errs, fl(o.By[rows]))
        between('%s acc_Wy (kink rows)' % tag, m.get_param('acc_Wy', (I, o.layers[-1]))[rows], o.acc['Wy'][rows], twin.acc['Wy'][rows], AR, AA, errs)
        between('%s acc_By (kink rows)' % tag, m.get_param('acc_By', (I,))[rows], o.acc['By'][rows], twin.acc['By'][rows], AR, AA, errs)
    if o.E is not None:
        upd('%s dE' % tag, m.get_param('E', (I, o.embedding)), o.E, init['E'])
        close_rel('%s acc_E' % tag, m.get_param('acc_E', (I, o.embedding)), o.acc['E'], AR, AA, errs)
    if o.momentum > 0:
        close_rel('%s vel_Wy' % tag, m.get_param('vel_Wy', (I, o.layers[-1])), o.vel['Wy'], PR, PA, errs)


def test_mfma_layout_selftest():
    assert _native.selftest_mfma() < 1e-5


def test_sample_store_bit_exact():
    o, m = make_pair(I=997, B=8, ns=64, store_rows=50, loss='bpr-max', constrained_embedding=True, layers=(12,))
    st = m.get_sample_store(64)
    np.testing.assert_array_equal(st, o.ST)


LOOSEN_60 = 20.0      # 60 steps on an 80-item catalogue: every row is rewritten dozens of times, rounding differences compound

CASES = {
    'bprmax_elu': dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(12,), bpreg=0.7),
    'bprmax_mom_drop': dict(loss='bpr-max', final_act='linear', constrained_embedding=True, layers=(16,), momentum=0.3,
                            dropout_p_hidden=0.3, dropout_p_embed=0.2),
    'top1max_2layer': dict(loss='top1-max', final_act='tanh', constrained_embedding=True, layers=(8, 12),
                           dropout_p_hidden=0.2),
    'xe_softmax_logq': dict(loss='cross-entropy', final_act='softmax', constrained_embedding=True, layers=(12,),
                            logq=1.0, sample_alpha=0.5, momentum=0.2),
    'xe_sep_embed': dict(loss='cross-entropy', final_act='softmax', constrained_embedding=False, embedding=20,
                         layers=(12,)),
    'bprmax_relu_sep2': dict(loss='bpr-max', final_act='relu', hidden_act='relu', constrained_embedding=False,
                             embedding=8, layers=(8, 8), lmbd=0.01),
    'bprmax_softmax': dict(loss='bpr-max', final_act='softmax', constrained_embedding=True, layers=(12,)),
    'bpr_linear': dict(loss='bpr', final_act='linear', constrained_embedding=True, layers=(12,)),
    'top1_tanh_mom': dict(loss='top1', final_act='tanh', constrained_embedding=True, layers=(12,), momentum=0.1),
    'xelogit_smooth': dict(loss='xe_logit', final_act='softmax_logit', constrained_embedding=True, layers=(12,),
                           smoothing=0.1, logq=1.0),
    'xe_smooth_sep': dict(loss='cross-entropy', final_act='softmax', constrained_embedding=False, embedding=8,
                          layers=(12,), smoothing=0.2),
    'xelogit_elu': dict(loss='xe_logit', final_act='elu-1.0', constrained_embedding=True, layers=(12,)),
    # generic optimizer path (raw gradients -> rule in the update kernels), incl. global-norm clipping
    'rmsprop_mom': dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(12,), adapt='rmsprop',
                        adapt_params=(0.9,), learning_rate=0.01, momentum=0.2),
    'adadelta_xe': dict(loss='cross-entropy', final_act='softmax', constrained_embedding=True, layers=(12,), adapt='adadelta',
                        adapt_params=(0.95,), learning_rate=1.0),
    'adam_sep': dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=False, embedding=8, layers=(12,), adapt='adam',
                     adapt_params=(0.9, 0.999), learning_rate=0.01),
    'sgd_cap_2layer': dict(loss='top1-max', final_act='tanh', constrained_embedding=True, layers=(8, 12), adapt=None,
                           grad_cap=0.05, learning_rate=0.05, momentum=0.1),
    'adagrad_cap_lmbd': dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(12,), grad_cap=0.02,
                             lmbd=0.01),
    'adam_onehot': dict(loss='cross-entropy', final_act='softmax', layers=(12,), adapt='adam', adapt_params=(0.9, 0.99),
                        learning_rate=0.01, dropout_p_hidden=0.2),
    # one-hot input (the reference's constructor default: no embedding, layer 0 reads rows of Wx[0])
    'onehot_bprmax': dict(loss='bpr-max', final_act='elu-0.5', layers=(12,), momentum=0.2),
    'onehot_xe_2layer': dict(loss='cross-entropy', final_act='softmax', layers=(8, 12), dropout_p_hidden=0.2, dropout_p_embed=0.3,
                             lmbd=0.01),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_first_step_intermediates(name):
    """One step: every intermediate the kernels expose against the oracle's debug dump."""
    kw = CASES[name]
    I, B, ns = 60, 20, 40
    o, m = make_pair(I, B, ns, store_rows=7, **kw)
    plan = random_plan(I, B, 1, seed=5)
    plan['M'][:] = B - 3          # exercise the inactive in-batch rows/columns as well
    m.set_plan(plan)
    M = int(plan['M'][0])
    cost, dbg = o.train_step(plan['in_idx'][0], plan['out_idx'][0], M, plan['reset'][0], return_debug=True)
    m.train_steps(0, 1)
    errs = []
    report('--- first step %s' % name)
    N = B + ns
    ld = int(m.get_debug('ldSc', (1,))[0])
    cols = np.r_[0:M, B:N]
    L = len(o.layers)
    for i, D in enumerate(o.layers):
        cch = dbg['caches'][i]
        close('hd%d' % i, m.get_debug('hd%d' % i, (B, D))[:M], cch['hd'], errs=errs)
        close('r%d' % i, m.get_debug('r%d' % i, (B, D))[:M], cch['r'], errs=errs)
        close('z%d' % i, m.get_debug('z%d' % i, (B, D))[:M], cch['z'], errs=errs)
        close('c%d' % i, m.get_debug('c%d' % i, (B, D))[:M], cch['c'], errs=errs)
    ds = m.get_debug('scores', (B, ld))
    close('ds', ds[:M][:, cols], dbg['ds'], atol=1e-6, rtol=1e-3, errs=errs)
    assert not ds[:M][:, M:B].any(), 'inactive in-batch columns must carry zero gradient'
    # the gradient producers store the per-occurrence Adagrad STEP lr * g / sqrt(acc_pre + g^2 + eps); acc_pre = 0 here
    generic = (o.adapt != 'adagrad' or o.grad_cap > 0)       # generic optimizer path: the producers leave raw gradients

    def step_of(g):
        g = np.asarray(g, dtype=np.float64)
        return g if generic else o.learning_rate * g / np.sqrt(g * g + 1e-6)
    close('dSy(step)', m.get_debug('dSy', (ld, o.layers[-1]))[cols], step_of(dbg['dSy']), atol=2e-6, rtol=2e-3, errs=errs)
    close('dSBy(step)', m.get_debug('dSBy', (ld,))[cols], step_of(dbg['dSBy']), atol=2e-6, rtol=2e-3, errs=errs)
    ks = int(m.get_debug('ksplit', (1,))[0])
    dhp = m.get_debug('dhpart', (ks, B, o.layers[-1])).sum(axis=0)
    close('dh_top', dhp[:M], dbg['dtop'], atol=1e-6, rtol=1e-3, errs=errs)
    n_in = o.layers[-1] if o.constrained_embedding else (o.embedding or 3 * o.layers[0])
    close('dSx(step)', m.get_debug('dSx', (B, n_in))[:M], step_of(dbg['dSx']), atol=2e-6, rtol=2e-3, errs=errs)
    close('cost', m.get_losses(0, 1), [cost], atol=2e-6, rtol=2e-4, errs=errs)
    compare_params(o, m, errs, 'p1', Mrows=M)
    assert not errs, errs


@pytest.mark.parametrize('name', sorted(CASES))
def test_loss_curve_and_weights_after_many_steps(name):
    """60 steps with a shrinking batch at the end and several sample-store wrap-arounds (refill path)."""
    kw = CASES[name]
    I, B, ns, T = 80, 12, 24, 60
    o, m = make_pair(I, B, ns, store_rows=9, **kw)
    plan = random_plan(I, B, T, seed=11, tail=True)
    m.set_plan(plan)
    want = []
    for t in range(T):
        M = int(plan['M'][t])
        want.append(o.train_step(plan['in_idx'][t], plan['out_idx'][t], M, plan['reset'][t]))
    m.train_steps(0, 25)
    m.train_steps(25, T - 25)
    got = m.get_losses(0, T)
    errs = []
    report('--- curve %s' % name)
    close('loss curve', got, np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    np.testing.assert_array_equal(m.get_sample_store(ns), o.ST)
    compare_params(o, m, errs, 'p60', Mrows=int(plan['M'][-1]), loosen=LOOSEN_60)
    assert not errs, errs


def test_graph_replay_is_bit_identical_to_eager():
    kw = CASES['bprmax_mom_drop']
    I, B, ns, T = 80, 12, 24, 80
    plan = random_plan(I, B, T, seed=13)
    outs = []
    for g in (0, 1):
        _, m = make_pair(I, B, ns, store_rows=200, use_graph=g, **dict(kw))
        m.set_plan(plan)
        m.train_steps(0, T)
        outs.append((m.get_losses(0, T), m.get_param('Wy', (I, 16)), m.get_param('Wx', (16, 48), 0)))
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_refill_boundaries_one_call_equals_stepwise():
    """Device-generated sample store of 3 rows: it is refilled every 3 steps (gru4rec.py:618-620).  One call over 20 steps
    (graph replay, the next step's inputs staged by the previous step's bookkeeping, re-staged after each refill) must equal
    20 one-step calls (every call stages its own inputs): same costs and weights to the last bit."""
    I, B, ns, T = 90, 12, 24, 20
    plan = random_plan(I, B, T, seed=41)
    rng = np.random.RandomState(5)
    pop = rng.randint(1, 30, size=I).astype(np.float64) ** 0.75
    cum = (pop.cumsum() / pop.sum()).astype(np.float32)
    cum[-1] = 1.0
    Wy = (rng.randn(I, 16) * 0.2).astype(np.float32)
    outs = []
    for mode in ('one_call', 'stepwise'):
        m = _native.Model(n_items=I, layers=[16], batch_size=B, n_sample=ns, loss=_native.LOSS_IDS['bpr-max'],
                          final_act=_native.ACT_IDS['elu'], final_act_p0=0.5, hidden_act=_native.ACT_IDS['tanh'], embed_mode=0,
                          learning_rate=0.1, momentum=0.2, bpreg=1.0, sample_alpha=0.75, dropout_p_hidden=0.1, sample_store=3 * ns,
                          seed=77, device=0, rank=0, nranks=1, use_graph=1 if mode == 'one_call' else 0)
        m.set_param('Wy', Wy)
        m.set_popularity(cum)
        assert m.sample_store_rows() == 3
        m.set_plan(plan)
        m.reset_hidden()
        if mode == 'one_call':
            m.train_steps(0, T)
        else:
            for t in range(T):
                m.train_steps(t, 1)
        outs.append((m.get_losses(0, T), m.get_param('Wy', (I, 16)), m.get_param('Wx', (16, 48), 0), m.get_sample_store(ns)))
        m.close()
    assert len(np.unique(outs[0][0])) == T          # every step saw different samples / inputs
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)


def test_multirank_data_path_on_one_gpu(monkeypatch):
    """The N > 1 step (dense gradients staged -> RCCL all-reduce on the side stream -> k_dense_apply, next to the
    sparse update) with a one-rank communicator must reproduce the fused single-GPU step."""
    kw = CASES['bprmax_mom_drop']
    I, B, ns, T = 80, 12, 24, 60
    plan = random_plan(I, B, T, seed=17)
    outs = []
    for staged in (0, 1):
        if staged:
            monkeypatch.setenv('G4R_FORCE_STAGED', '1')
        _, m = make_pair(I, B, ns, store_rows=200, use_graph=0, **dict(kw))
        if staged:
            m.comm_init(_native.comm_unique_id(), 1, 0)
            assert m.comm_min(T) == T
        m.set_plan(plan)
        m.train_steps(0, T)
        if staged:
            m.comm_sync_sparse()
        outs.append((m.get_losses(0, T), m.get_param('Wy', (I, 16)), m.get_param('Wx', (16, 48), 0),
                     m.get_param('Bh', (48,), 0), m.get_param('acc_Wh', (16, 16), 0)))
        m.close()
    # (the two paths run different dense-gradient tiles -- k_update_l's 16 x 64 register-fed tiles against k_dense_grad's LDS-staged
    # 32 x 32 ones since round 6 --, i.e. different fp32 summation orders over the batch: equal to rounding, not to the bit)
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6)


def test_epoch_with_real_schedule_and_compaction():
    """Sessions -> C++ plan -> device epoch (with H compaction at the tail) vs the oracle driven by the
    literal restatement of the reference loop."""
    rng = np.random.RandomState(7)
    n_sess, I, B, ns = 90, 70, 16, 24
    lens = rng.randint(1, 8, size=n_sess)
    off = np.zeros(n_sess + 1, dtype=np.int32)
    off[1:] = np.cumsum(lens)
    items = rng.randint(0, I, size=off[-1]).astype(np.int32)
    order = rng.permutation(n_sess)
    o, m = make_pair(I, B, ns, store_rows=5, loss='bpr-max', final_act='elu-1', constrained_embedding=True,
                     layers=(12,), momentum=0.1)
    for i in range(len(o.layers)):
        o.H[i][:] = 0
    m.reset_hidden()
    plan = _native.build_plan(off, order, items, B, ns)
    m.set_plan(plan)
    want = []
    for ev in fit_schedule(off, order, items, B, ns):
        if ev[0] == 'step':
            want.append(o.train_step(ev[1], ev[2], ev[3], ev[4]))
        else:
            valid = ev[1]
            for i in range(len(o.layers)):
                H = np.zeros_like(o.H[i])
                keep = o.H[i][:len(valid)][valid]
                H[:len(keep)] = keep
                o.H[i] = H
    assert plan['T'] == len(want) and plan['n_compact'] > 0
    m.train_steps(0, plan['T'])
    errs = []
    report('--- epoch with compaction')
    close('loss curve', m.get_losses(0, plan['T']), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    Mlast = int(plan['M'][-1])
    close('H tail', m.get_param('H', (B, 12), 0)[:Mlast], o.H[0][:Mlast], errs=errs)
    close('Wy', m.get_param('Wy', (I, 12)), o.Wy, 1e-4, 2e-3, errs=errs)
    assert not errs, errs


def test_baseline_config2_shape_few_steps():
    """BASELINE config #2 shape: I = 37,483, B = 128, layers = [100], n_sample = 2048, BPR-max, constrained."""
    I, B, ns, T = 37483, 128, 2048, 6
    o, m = make_pair(I, B, ns, store_rows=16, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True,
                     layers=(100,), learning_rate=0.1, bpreg=1.0)
    plan = random_plan(I, B, T, seed=17)
    # make duplicates certain: popular items repeated inside the batch and against the samples
    plan['in_idx'][:, :8] = o.ST[0][:8]
    plan['out_idx'][:, 8:16] = plan['in_idx'][:, :8]
    m.set_plan(plan)
    twin = kink_twin(o)
    want, kink = oracle_steps(o, plan, T, full_batch=B, twin=twin)
    m.train_steps(0, T)
    errs = []
    report('--- config #2 shape (%d items with a score on the elu kink: bounded by the two slopes)' % len(kink))
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    # the items on the kink: every element of their rows between the oracle's two slopes (a factor 1 / alpha = 2 on one element of
    # 128 x 2176), every other row against the oracle as usual
    compare_params(o, m, errs, 'cfg2', skip_items=kink, twin=twin)
    assert len(kink) <= 8, len(kink)      # (a handful per million scores)
    assert not errs, errs


def test_wide_layer_and_big_batch():
    """D = 320 with B = 160 (two 128-row blocks, K-chunked scoring) and a 2-chunk sparse row."""
    I, B, ns, T = 3000, 160, 512, 3
    o, m = make_pair(I, B, ns, store_rows=8, loss='cross-entropy', final_act='softmax', constrained_embedding=True,
                     layers=(320,), learning_rate=0.05, logq=1.0, dropout_p_embed=0.3)
    plan = random_plan(I, B, T, seed=19)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], B, plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- wide layer')
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, 'wide')
    assert not errs, errs


def test_many_negatives_big_batch():
    """B = 300, 4200 negatives: the large-shape variants (64-deep K chunks in the scoring kernel, split-K slabs wider than
    one chunk in its backward, 64-column GRU tiles at D = 256) against the oracle."""
    I, B, ns, T = 6000, 300, 4200, 2
    o, m = make_pair(I, B, ns, store_rows=4, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True,
                     layers=(256,), learning_rate=0.05, bpreg=0.5)
    plan = random_plan(I, B, T, seed=29)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], B, plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- many negatives')
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, 'manyneg')
    assert not errs, errs


def test_more_split_k_slabs_than_one_batch():
    """B + n_sample = 1632 -> 13 score chunks of 128 -> 13 split-K slabs of dh: k_gru_bwd_fused sums them in two batches
    (its first batch holds 10), in the same fixed order."""
    I, B, ns, T = 900, 32, 1600, 3
    o, m = make_pair(I, B, ns, store_rows=5, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(24,),
                     learning_rate=0.05, momentum=0.1, dropout_p_hidden=0.2)
    assert int(m.get_debug('ksplit', (1,))[0]) == 13
    plan = random_plan(I, B, T, seed=31)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], B, plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- 13 slabs')
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, 'slabs13')
    assert not errs, errs


@pytest.mark.parametrize('name', ['bprmax_elu', 'xe_softmax_logq', 'top1max_2layer', 'xe_sep_embed', 'xelogit_smooth', 'onehot_bprmax'])
def test_predict_and_ranks(name):
    kw = CASES[name]
    I, B = 300, 24
    o, m = make_pair(I, B, 0, store_rows=0, **kw)
    rng = np.random.RandomState(23)
    pb = 40
    m.predict_begin(pb)
    H = [np.zeros((pb, D), dtype=np.float32) for D in o.layers]
    errs = []
    report('--- predict %s' % name)
    for step in range(3):
        in_idx = rng.randint(0, I, size=pb)
        tgt = rng.randint(0, I, size=pb)
        want, H = o.predict_step(H, in_idx)
        got = m.predict_step(in_idx)
        close('scores step %d' % step, got, want, atol=2e-6, rtol=3e-4, errs=errs)
        for mode in ('standard', 'conservative', 'median'):
            r = m.rank_targets(tgt, 0, mode)
            t = got[np.arange(pb), tgt][:, None]
            gt = (got > t).sum(axis=1)
            eq = (got == t).sum(axis=1)
            ref = {'standard': gt + 1, 'conservative': gt + eq, 'median': gt + 0.5 * (eq - 1) + 1}[mode]
            np.testing.assert_array_equal(r, ref.astype(np.float32))
    # subset of items + hidden-state maintenance
    sel = rng.permutation(I)[:77]
    in_idx = rng.randint(0, I, size=pb)
    zero = (rng.rand(pb) < 0.3)
    for i in range(len(H)):
        H[i] = H[i].copy()
        H[i][zero] = 0
    m.predict_hidden(zero_mask=zero.astype(np.uint8))
    want, H = o.predict_step(H, in_idx, sel)
    got = m.predict_step(in_idx, sel)
    close('scores subset', got, want, atol=2e-6, rtol=3e-4, errs=errs)
    keep = np.sort(rng.permutation(pb)[:31]).astype(np.int32)
    m.predict_hidden(keep_rows=keep)
    H = [h[keep] for h in H]
    in_idx = rng.randint(0, I, size=len(keep))
    want, H = o.predict_step(H, in_idx)
    got = m.predict_step(in_idx)
    close('scores after compaction', got, want, atol=2e-6, rtol=3e-4, errs=errs)
    assert not errs, errs


@pytest.mark.parametrize('fa', ['elu-0.5', 'leaky-0.2', 'selu-1.05-1.67'])
def test_activation_derivative_branch_follows_the_input_sign(fa):
    """Scores a rounding away from 0 (|s| ~ 1e-9: exp(s) - 1 rounds to 0 in fp32): the reference's T.grad switches the piecewise
    activations on the INPUT (gru4rec.py:214-218: T.switch(T.ge(X, 0), ...)), so d cost / d s of a tiny negative score carries
    alpha, not 1.  (The kernels only keep the output; the branch travels in the sign bit of the zero.)"""
    I, B, ns = 60, 20, 40
    o, m = make_pair(I, B, ns, store_rows=7, loss='bpr-max', final_act=fa, constrained_embedding=False, embedding=8, layers=(12,))
    tiny = np.where(np.arange(I)[:, None] % 2 == 0, 1e-9, -1e-9).astype(np.float32) * np.ones((1, 12), dtype=np.float32)
    o.Wy[:40] = tiny[:40]
    o.By[:] = 0
    m.set_param('Wy', o.Wy)
    m.set_param('By', o.By)
    plan = random_plan(I, B, 1, seed=5)
    m.set_plan(plan)
    cost, dbg = o.train_step(plan['in_idx'][0], plan['out_idx'][0], B, plan['reset'][0], return_debug=True)
    m.train_steps(0, 1)
    ld = int(m.get_debug('ldSc', (1,))[0])
    ds = m.get_debug('scores', (B, ld))[:, :B + ns]
    want = dbg['ds']
    assert (np.abs(want) > 0).all()
    np.testing.assert_allclose(ds, want, rtol=2e-3, atol=1e-9)
    close('cost', m.get_losses(0, 1), [cost], atol=2e-6, rtol=2e-4)
