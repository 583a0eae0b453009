"""Randomised shape sweep on the MI355X: batch sizes that are not tile multiples, layer widths that are only multiples
of 4, 0 / few / many sampled negatives, 1-3 layers of different widths, all three input modes and all losses -- three
steps each against the fp32 oracle (costs and touched parameters)."""
import numpy as np
import pytest

from gru4rec_amd import _native
from oracle.model import OracleGRU4Rec, parse_act

pytestmark = pytest.mark.gpu

LOSSES = [('bpr-max', 'elu-0.5'), ('top1-max', 'tanh'), ('cross-entropy', 'softmax'), ('bpr', 'linear'), ('top1', 'tanh'),
          ('xe_logit', 'softmax_logit'), ('bpr-max', 'softmax'), ('cross-entropy', 'softmax')]


def make_cases(n=36, seed=2024):
    rng = np.random.RandomState(seed)
    cases = []
    for i in range(n):
        nl = int(rng.choice([1, 1, 2, 3]))
        layers = tuple(int(4 * rng.randint(2, 70)) for _ in range(nl))
        mode = int(rng.choice([0, 0, 1, 2]))
        if mode == 2 and 3 * layers[0] > 512:
            layers = (int(4 * rng.randint(2, 42)),) + layers[1:]
        loss, fa = LOSSES[i % len(LOSSES)]
        cases.append(dict(layers=layers, B=int(rng.choice([1, 3, 17, 31, 33, 48, 80, 100, 144, 200])),
                          ns=int(rng.choice([0, 5, 64, 333, 2048])), mode=mode, emb=int(4 * rng.randint(1, 40)),
                          loss=loss, final_act=fa, momentum=float(rng.choice([0.0, 0.3])),
                          dh=float(rng.choice([0.0, 0.25])), de=float(rng.choice([0.0, 0.3])),
                          logq=float(rng.choice([0.0, 1.0])) if loss == 'cross-entropy' else 0.0, seed=int(rng.randint(1 << 30))))
    return cases


CASES = make_cases()


@pytest.mark.parametrize('ci', range(len(CASES)))
def test_shape(ci):
    c = CASES[ci]
    I, T, rows = 700, 3, 5
    B, ns = c['B'], c['ns']
    if ns == 0 and B < 2:
        B = 2                        # in-batch negatives need at least one other row (gru4rec.py:637)
    kw = dict(n_items=I, layers=c['layers'], batch_size=B, loss=c['loss'], final_act=c['final_act'], n_sample=ns,
              sample_alpha=0.6, learning_rate=0.1, momentum=c['momentum'], bpreg=0.8, logq=c['logq'],
              dropout_p_hidden=c['dh'], dropout_p_embed=c['de'], constrained_embedding=(c['mode'] == 0),
              embedding=(c['emb'] if c['mode'] == 1 else 0), dtype=np.float32, seed=c['seed'] % 100000)
    o = OracleGRU4Rec(**kw)
    rng = np.random.RandomState(c['seed'])
    o.set_popularity(rng.randint(1, 50, size=I))
    o.make_sample_store(rows * ns if ns else 0)
    fa = parse_act(c['final_act'])
    m = _native.Model(n_items=I, layers=list(c['layers']), batch_size=B, n_sample=ns, loss=_native.LOSS_IDS[c['loss']],
                      final_act=_native.ACT_IDS[fa[0]], final_act_p0=fa[1], final_act_p1=fa[2], hidden_act=_native.ACT_IDS['tanh'],
                      embed_mode=c['mode'], embedding=kw['embedding'], learning_rate=0.1, momentum=c['momentum'], lmbd=0.0,
                      bpreg=0.8, logq=c['logq'], sample_alpha=0.6, dropout_p_hidden=c['dh'], dropout_p_embed=c['de'],
                      sample_store=rows * ns if ns else 0, seed=kw['seed'], device=0, rank=0, nranks=1, use_graph=0)
    for i, D in enumerate(o.layers):
        o.Bh[i] = (rng.randn(3 * D) * 0.1).astype(np.float32)
        o.H[i] = (rng.randn(B, D) * 0.3).astype(np.float32)
        for n in ('Wx', 'Wh', 'Wrz', 'Bh', 'H'):
            m.set_param(n, getattr(o, n)[i], i)
    m.set_param('Wy', o.Wy)
    m.set_param('By', o.By)
    if o.E is not None:
        m.set_param('E', o.E)
    m.set_popularity(o.P, o.lq_tgt if o.logq else None, o.lq_smp if o.logq else None)
    M = np.array([B, max(1, B - 2) if ns else max(2, B - 2), max(1, B // 2) if ns else max(2, B // 2)], dtype=np.int32)
    plan = dict(in_idx=rng.randint(0, I, size=(T, B)).astype(np.int32), out_idx=rng.randint(0, I, size=(T, B)).astype(np.int32),
                reset=(rng.rand(T, B) < 0.3).astype(np.uint8), M=M, T=T, n_compact=0)
    m.set_plan(plan)
    want = [o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(M[t]), plan['reset'][t].astype(bool)) for t in range(T)]
    m.train_steps(0, T)
    got = m.get_losses(0, T)
    assert np.isfinite(got).all(), (c, got)
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-5, err_msg=str(c))
    # Parameters after three Adagrad steps.  The first steps divide by sqrt(g^2 + 1e-6): where |g| ~ 1e-3 a relative
    # error of 1e-2 in g (sums of a few hundred hardware-sigmoid terms that cancel) moves the step by ~1e-3, so a few
    # elements in 1e5 may sit outside the element-wise tolerance; they must stay rare and small.
    def check(name, got_p, want_p):
        err = np.abs(got_p.astype(np.float64) - want_p)
        bad = err > 3e-4 + 5e-3 * np.abs(want_p)
        assert bad.mean() < 1e-4 and err.max() < 5e-3, (name, c, int(bad.sum()), float(err.max()))
    check('Wy', m.get_param('Wy', (I, c['layers'][-1])), o.Wy)
    D0 = c['layers'][0]
    check('Wrz0', m.get_param('Wrz', (D0, 2 * D0), 0), o.Wrz[0])
    m.close()


PRED_CASES = [dict(I=53, pb=1, layers=(8,), mode=0), dict(I=700, pb=5, layers=(100,), mode=0), dict(I=3001, pb=37, layers=(36, 132), mode=0),
              dict(I=1000, pb=130, layers=(260,), mode=1, emb=44), dict(I=515, pb=100, layers=(64,), mode=2),
              dict(I=4099, pb=512, layers=(480,), mode=0), dict(I=129, pb=33, layers=(12, 20, 28), mode=1, emb=4)]


@pytest.mark.parametrize('ci', range(len(PRED_CASES)))
def test_predict_shape(ci):
    """predict_next_batch's device path at odd batch / catalogue / layer sizes: scores vs the oracle, 3 steps + a subset."""
    c = PRED_CASES[ci]
    I, pb = c['I'], c['pb']
    o = OracleGRU4Rec(n_items=I, layers=c['layers'], batch_size=8, loss='cross-entropy', final_act='softmax', n_sample=0,
                      constrained_embedding=(c['mode'] == 0), embedding=c.get('emb', 0) if c['mode'] == 1 else 0,
                      dtype=np.float32, seed=3)
    m = _native.Model(n_items=I, layers=list(c['layers']), batch_size=8, n_sample=0, loss=0, final_act=_native.ACT_IDS['softmax'],
                      hidden_act=_native.ACT_IDS['tanh'], embed_mode=c['mode'], embedding=o.embedding or 0, learning_rate=0.1,
                      sample_store=0, seed=3, device=0, rank=0, nranks=1, use_graph=0)
    rng = np.random.RandomState(ci)
    for i, D in enumerate(o.layers):
        o.Bh[i] = (rng.randn(3 * D) * 0.1).astype(np.float32)
        for n in ('Wx', 'Wh', 'Wrz', 'Bh'):
            m.set_param(n, getattr(o, n)[i], i)
    o.By = (rng.randn(I) * 0.2).astype(np.float32)
    m.set_param('Wy', o.Wy)
    m.set_param('By', o.By)
    if o.E is not None:
        m.set_param('E', o.E)
    m.predict_begin(pb)
    H = [np.zeros((pb, D), dtype=np.float32) for D in o.layers]
    for step in range(3):
        in_idx = rng.randint(0, I, size=pb)
        want, H = o.predict_step(H, in_idx)
        got = m.predict_step(in_idx)
        np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-6, err_msg=str(c))
        tgt = rng.randint(0, I, size=pb)
        r = m.rank_targets(tgt, 0, 'standard')
        t = got[np.arange(pb), tgt][:, None]
        np.testing.assert_array_equal(r, ((got > t).sum(axis=1) + 1).astype(np.float32))
    sel = rng.permutation(I)[:max(1, I // 3)]
    in_idx = rng.randint(0, I, size=pb)
    want, H = o.predict_step(H, in_idx, sel)
    got = m.predict_step(in_idx, sel)
    np.testing.assert_allclose(got, want, rtol=1e-3, atol=1e-6, err_msg=str(c))
    m.close()


def test_large_catalogue_properties():
    """BASELINE configs[2] scale (Rees46 shape: D = 512, batch 240, 2048 negatives; 1 M items here) where the oracle is too slow:
    size-independent properties instead -- graph replay == eager launches bit for bit, rows of items that never occur keep
    their initial bits (parameters AND accumulators), input / target rows all move, costs finite and decreasing on a repeated batch."""
    I, D, B, ns, T = 1000003, 512, 240, 2048, 40
    rng = np.random.RandomState(0)
    Wy = (rng.rand(I, D).astype(np.float32) - 0.5) * 0.05
    support = np.maximum(1, (1e6 / (1 + np.arange(I))).astype(np.int64))
    pop = support.astype(np.float64) ** 0.5
    P = (pop.cumsum() / pop.sum()).astype(np.float32)
    P[-1] = 1
    dense = {n: (rng.rand(*shape).astype(np.float32) - 0.5) * 0.1 for n, shape in
             (('Wx', (D, 3 * D)), ('Wh', (D, D)), ('Wrz', (D, 2 * D)))}
    base = rng.randint(0, I, size=(2, B)).astype(np.int32)
    plan = dict(in_idx=np.tile(base[0], (T, 1)), out_idx=np.tile(base[1], (T, 1)), reset=np.zeros((T, B), dtype=np.uint8),
                M=np.full(T, B, dtype=np.int32), T=T, n_compact=0)
    outs = []
    for use_graph in (0, 1):
        m = _native.Model(n_items=I, layers=[D], batch_size=B, n_sample=ns, loss=_native.LOSS_IDS['cross-entropy'],
                          final_act=_native.ACT_IDS['softmax'], hidden_act=_native.ACT_IDS['tanh'], embed_mode=0, embedding=0,
                          learning_rate=0.065, momentum=0.0, logq=1.0, sample_alpha=0.5, dropout_p_embed=0.45,
                          sample_store=ns * 64, seed=5, device=0, rank=0, nranks=1, use_graph=use_graph)
        for n, a in dense.items():
            m.set_param(n, a, 0)
        m.set_param('Wy', Wy)
        lq_t = np.log(support.astype(np.float32))
        m.set_popularity(P, lq_t, np.log(support.astype(np.float32) ** np.float32(0.5)))
        m.set_plan(plan)
        m.train_steps(0, T)
        store = m.get_sample_store(ns)
        outs.append((m.get_losses(0, T), m.get_param('Wy', (I, D)), m.get_param('acc_Wy', (I, D)), m.get_param('Wh', (D, D), 0), store))
        m.close()
    for a, b in zip(outs[0], outs[1]):
        np.testing.assert_array_equal(a, b)                       # graph replay == eager, bit for bit
    losses, Wy2, acc, _, store = outs[0]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]
    touched = np.zeros(I, dtype=bool)
    touched[base.ravel()] = True
    touched[store[:T].ravel()] = True
    changed = (Wy2 != Wy).any(axis=1)
    assert not changed[~touched].any() and not acc[~touched].any()     # untouched rows keep their bits
    # every input / target row moved; sampled negatives move unless their softmax weight has underflowed to ~0 (the 40
    # repeats of one batch make the model confident), and whatever gradient they saw is in the accumulator
    never = touched & ~changed
    assert not np.isin(np.nonzero(never)[0], base.ravel()).any()
    assert never.sum() < 0.2 * touched.sum() and (acc[touched] >= 0).all() and (acc[touched & changed] > 0).any(axis=1).all()
