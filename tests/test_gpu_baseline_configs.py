"""Oracle parity at the EXACT shapes of BASELINE.json configs[1..4] (catalogue reduced where the oracle would not fit in
seconds: the item count changes neither a tile shape nor a code path) and on bench.py's own workload.

Tolerances (fp32 HIP path vs the NumPy oracle; both fp32, different summation orders):
  per-step cost   rtol 5e-4 + atol 5e-6
  parameters as updates (value - initial value) and accumulators against their own scale: test_gpu_parity.compare_params
  (1e-3 |update| + 1e-4 max|update|; 2e-4 |acc| + 1e-5 max|acc|) -- a 1 % error of an accumulator or of a step fails.
The long bench-plan run brackets its tolerance by the oracle's own fp32-vs-fp64 gap (stated in the test)."""
import numpy as np
import pytest

import bench
from gru4rec_amd import _native
from oracle.model import OracleGRU4Rec

from test_gpu_parity import close, compare_params, kink_twin, make_pair, oracle_steps, random_plan, report, snapshot

pytestmark = pytest.mark.gpu


def _run(tag, I, B, ns, T, store_rows, dup=True, **kw):
    o, m = make_pair(I, B, ns, store_rows=store_rows, **kw)
    plan = random_plan(I, B, T, seed=101)
    if dup:      # repeated items inside the batch and against the negatives (duplicate semantics of the sparse update)
        plan['in_idx'][:, :8] = o.ST[0][:8] if ns else plan['out_idx'][:, 24:32]
        plan['out_idx'][:, 8:16] = plan['in_idx'][:, :8]
        plan['out_idx'][:, 16:20] = plan['out_idx'][:, 20:24]
    m.set_plan(plan)
    twin = kink_twin(o)      # (the oracle on the other slope of a piecewise final activation, for the scores on its kink)
    want, kink = oracle_steps(o, plan, T, full_batch=B, twin=twin)
    m.train_steps(0, T)
    errs = []
    report('--- %s (%d kink items: bounded by the two slopes)' % (tag, len(kink)))
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    if ns:
        np.testing.assert_array_equal(m.get_sample_store(ns), o.ST)
    compare_params(o, m, errs, tag, skip_items=kink, twin=twin)
    m.close()
    n_scores = float(T) * B * (B + ns)
    assert len(kink) <= max(4, int(20e-6 * n_scores)), (len(kink), n_scores)      # (a handful per million scores)
    assert not errs, errs


def test_cfg1_exact_shape():
    """configs[0] (the reference's CPU-runnable plumbing case): layers=[100], batch=32, n_sample=0 -- in-batch negatives only --,
    cross-entropy over softmax, the RSC15 catalogue of 37,483 items; 16 steps, with a tail of shrinking batches (M < B: the
    reference's loop ends an epoch that way, gru4rec.py:647-651)."""
    I, B, T = 37483, 32, 16
    o, m = make_pair(I, B, 0, store_rows=0, loss='cross-entropy', final_act='softmax', constrained_embedding=True, layers=(100,),
                     learning_rate=0.1)
    plan = random_plan(I, B, T, seed=101, tail=True)
    plan['in_idx'][:, :8] = plan['out_idx'][:, 24:32]      # inputs that are other rows' targets, repeated targets
    plan['out_idx'][:, 8:12] = plan['out_idx'][:, 12:16]
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t]) for t in range(T)]
    m.train_steps(0, T)
    errs = []
    report('--- cfg1')
    close('loss curve', m.get_losses(0, T), np.array(want), atol=5e-6, rtol=5e-4, errs=errs)
    compare_params(o, m, errs, 'cfg1', Mrows=int(plan['M'].min()), loosen=2.0)
    m.close()
    assert not errs, errs


def test_cfg4_exact_shape():
    """configs[3]: layers=[256], batch=512, n_sample=8192, BPR-max (per-GPU shape of the 8-GPU run); 10 steps."""
    _run('cfg4', I=20000, B=512, ns=8192, T=10, store_rows=12, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True,
         layers=(256,), learning_rate=0.1, bpreg=1.0)


def test_cfg5_exact_shape():
    """configs[4]: layers=[100,100], TOP1-max, embedding dropout, batch=128, n_sample=2048; 12 steps."""
    _run('cfg5', I=37483, B=128, ns=2048, T=12, store_rows=16, loss='top1-max', final_act='elu-0.5', constrained_embedding=True,
         layers=(100, 100), learning_rate=0.1, dropout_p_embed=0.2)


def test_cfg3_exact_shape():
    """configs[2]: layers=[512], batch=240, n_sample=2048, cross-entropy + logQ, embedding dropout 0.45; 10 steps."""
    _run('cfg3', I=30000, B=240, ns=2048, T=10, store_rows=12, loss='cross-entropy', final_act='softmax', constrained_embedding=True,
         layers=(512,), learning_rate=0.065, logq=1.0, sample_alpha=0.5, dropout_p_embed=0.45)


def test_cfg2_exact_shape_with_momentum():
    """configs[1] shape with the momentum variant of the sparse update (velocity rows) switched on; 12 steps."""
    _run('cfg2mom', I=37483, B=128, ns=2048, T=12, store_rows=16, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True,
         layers=(100,), learning_rate=0.1, bpreg=1.0, momentum=0.1)


LOOSEN_240 = 40.0


def _bench_pair(cfg, steps, store_rows):
    plan, support = bench.make_plan(cfg, steps, 0, 1)
    m = bench.create_model(cfg, support, 0, 1, 0, None, use_graph=True, sample_store=store_rows * cfg['n_sample'])
    os_ = {}
    for dt in (np.float32, np.float64):
        o = OracleGRU4Rec(n_items=cfg['n_items'], layers=tuple(cfg['layers']), batch_size=cfg['batch_size'], loss=cfg['loss'],
                          final_act=cfg['final_act'], n_sample=cfg['n_sample'], sample_alpha=cfg['sample_alpha'],
                          learning_rate=cfg['learning_rate'], momentum=cfg['momentum'], bpreg=cfg['bpreg'], logq=cfg['logq'],
                          dropout_p_hidden=cfg['dropout_p_hidden'], dropout_p_embed=cfg['dropout_p_embed'],
                          constrained_embedding=True, dtype=dt, seed=12345)
        for i, D in enumerate(cfg['layers']):      # the bench's own initial weights
            n_in = cfg['layers'][i - 1] if i else cfg['layers'][-1]
            o.Wx[i] = m.get_param('Wx', (n_in, 3 * D), i).astype(dt)
            o.Wh[i] = m.get_param('Wh', (D, D), i).astype(dt)
            o.Wrz[i] = m.get_param('Wrz', (D, 2 * D), i).astype(dt)
        o.Wy = m.get_param('Wy', (cfg['n_items'], cfg['layers'][-1])).astype(dt)
        o.set_popularity(support)
        o.make_sample_store(store_rows * cfg['n_sample'])
        o.init0 = snapshot(o)
        os_[dt] = o
    return plan, m, os_


def test_bench_plan_loss_curve_240_steps():
    """bench.py's cfg2 workload (its generator, its plan, its initial weights) for 240 steps, graph replay, against the fp32
    oracle.  Tolerance: rtol 5e-4 widened, step by step, by the oracle's own fp32-vs-fp64 disagreement on the same plan
    (the two oracles differ only in rounding, so their gap is what "agreement" can mean at that step).  The cost must also
    stay in the regime RSC15 runs are in (0.69 -> ~0.6, no excursions): the round-1 generator's 9 % head item made it
    jump to 1e3 at steps 9-11."""
    cfg = bench.CONFIGS['cfg2']
    T = 240
    plan, m, os_ = _bench_pair(cfg, T + 8, store_rows=256)
    for k in ('in_idx', 'out_idx', 'reset', 'M'):
        plan[k] = plan[k][:T]
    plan['T'], plan['n_compact'] = T, 0
    m.set_plan(plan)
    m.reset_hidden()
    m.train_steps(0, T)
    got = m.get_losses(0, T).astype(np.float64)
    c32 = np.array([os_[np.float32].train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t]) for t in range(T)], dtype=np.float64)
    c64 = np.array([os_[np.float64].train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t]) for t in range(T)], dtype=np.float64)
    gap = np.abs(c32 - c64)
    tol = 5e-6 + 5e-4 * np.abs(c32) + 4.0 * gap
    err = np.abs(got - c32)
    report('--- bench plan 240 steps: max err %.3e, max fp32/fp64 oracle gap %.3e, cost %.4f -> %.4f, max %.4f' % (
        err.max(), gap.max(), c32[0], c32[-1], c32.max()))
    assert np.isfinite(got).all()
    assert (err <= tol).all(), (int(np.argmax(err / tol)), float((err / tol).max()))
    assert c32.max() < 0.75 and c32[-1] < c32[0]      # a sane curve: no blow-up inside bench.py's timed window
    errs = []
    # 240 steps: the oracle's own fp32-vs-fp64 gap on Wy grows to ~1e-4 of an update over such a run (see the module docstring of
    # test_gpu_e2e_recall.py); the bounds are widened accordingly
    compare_params(os_[np.float32], m, errs, 'bench240', loosen=LOOSEN_240)
    m.close()
    assert not errs, errs
