"""Deferred row updates (g4r_update_kernels.cuh: k_defer_scan / k_sparse_flush): a row whose item is not gathered again before the end of
the current window of steps (one replay of the step graph) is applied by the window's flush launch instead of by its step's update
launch.  The claim is EXACTNESS: same operands, same arithmetic, so every loss, parameter and accumulator has the same BITS as with
G4R_DEFER=0 on the same update kernel (the deferred mode runs the merged k_update; where the immediate mode would take k_update_l, whose
dense tiles add the batch in another order, the reference run sets G4R_LEAN_UPDATE=0) -- across windows, call boundaries, sample-store refills inside a call, batch tails, catalogues small enough that most
items are gathered again right away, two item tables, one-hot input, two layers.  (Integer-exact comparison: assert_array_equal.)"""
import os

import numpy as np
import pytest

from test_gpu_parity import make_pair, random_plan

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _merged_update_in_both_runs(monkeypatch):
    monkeypatch.setenv('G4R_LEAN_UPDATE', '0')      # read by g4r_create: immediate and deferred runs on the merged k_update

CASES = {
    # name: (I, B, ns, T, store_rows, calls, kwargs)
    'cfg2_shape': (37483, 128, 2048, 70, 40, (70,), dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(100,),
                                                          learning_rate=0.1, bpreg=1.0)),
    'tiny_catalogue_everything_repeats': (300, 32, 64, 75, 200, (37, 38), dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True,
                                                                                layers=(32,), learning_rate=0.1, bpreg=1.0)),
    'refills_inside_the_call': (5000, 64, 256, 100, 21, (100,), dict(loss='cross-entropy', final_act='softmax', constrained_embedding=True,
                                                                     layers=(64,), learning_rate=0.07, logq=1.0, dropout_p_embed=0.2)),
    'separate_embedding_two_tables': (4000, 48, 128, 52, 60, (20, 32), dict(loss='top1-max', final_act='elu-0.5', constrained_embedding=False,
                                                                            embedding=24, layers=(40,), learning_rate=0.1)),
    'one_hot_input': (900, 32, 64, 40, 50, (40,), dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=False, embedding=0,
                                                        layers=(48,), learning_rate=0.1, bpreg=0.5)),
    'two_layers_wide_rows': (6000, 96, 512, 36, 40, (36,), dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(128, 320),
                                                                learning_rate=0.1, bpreg=1.0, dropout_p_hidden=0.1)),
}


def _run(case, defer):
    I, B, ns, T, store_rows, calls, kw = CASES[case]
    old = os.environ.get('G4R_DEFER')
    os.environ['G4R_DEFER'] = '1' if defer else '0'
    try:
        o, m = make_pair(I, B, ns, store_rows=store_rows, use_graph=1, **dict(kw))
    finally:
        if old is None:
            os.environ.pop('G4R_DEFER', None)
        else:
            os.environ['G4R_DEFER'] = old
    plan = random_plan(I, B, T, seed=5, tail=True)
    # sessions as the reference's loop makes them: a row's next input is its current target unless the session ends
    for t in range(1, T):
        keep = plan['reset'][t - 1] == 0
        plan['in_idx'][t][keep] = plan['out_idx'][t - 1][keep]
    m.set_plan(plan)
    t0 = 0
    for n in calls:
        m.train_steps(t0, n)
        t0 += n
    D = kw['layers'][-1]
    out = {'loss': m.get_losses(0, T).copy(), 'Wy': m.get_param('Wy', (I, D)).copy(), 'By': m.get_param('By', (I,)).copy(),
           'acc_Wy': m.get_param('acc_Wy', (I, D)).copy(), 'acc_By': m.get_param('acc_By', (I,)).copy()}
    for i, Dl in enumerate(kw['layers']):
        out['Wh%d' % i] = m.get_param('Wh', (Dl, Dl), i).copy()
        out['Bh%d' % i] = m.get_param('Bh', (3 * Dl,), i).copy()
    if not kw['constrained_embedding'] and kw.get('embedding'):
        out['E'] = m.get_param('E', (I, kw['embedding'])).copy()
        out['acc_E'] = m.get_param('acc_E', (I, kw['embedding'])).copy()
    if not kw['constrained_embedding'] and not kw.get('embedding'):
        out['Wx0'] = m.get_param('Wx', (I, 3 * kw['layers'][0]), 0).copy()
    stats = m.get_debug('defer_stats', 4)
    m.close()
    return out, stats


@pytest.mark.parametrize('case', sorted(CASES))
def test_deferred_updates_leave_identical_bits(case):
    ref, st0 = _run(case, defer=False)
    got, st1 = _run(case, defer=True)
    assert st0[2] == 0 and st1[2] == 1, (st0, st1)
    assert st1[0] > 0, 'no row update was deferred: the test would prove nothing (%s)' % (st1,)
    assert np.isfinite(ref['loss']).all()
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k], err_msg=k)
    print('%s: %d row updates (%d with a bias entry) went through flush launches' % (case, int(st1[0]), int(st1[1])))


def test_deferred_updates_without_the_step_graph():
    """Eager launches (use_graph = 0; what the counter passes of tools/final_profile.sh run): the same windows, the same bits."""
    I, B, ns, T = 3000, 48, 128, 40
    kw = dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(48,), learning_rate=0.1, bpreg=1.0)
    outs = []
    for defer, graph in ((0, 0), (1, 0), (1, 1)):
        old = os.environ.get('G4R_DEFER')
        os.environ['G4R_DEFER'] = str(defer)
        try:
            o, m = make_pair(I, B, ns, store_rows=50, use_graph=graph, **kw)
        finally:
            if old is None:
                os.environ.pop('G4R_DEFER', None)
            else:
                os.environ['G4R_DEFER'] = old
        plan = random_plan(I, B, T, seed=9, tail=True)
        m.set_plan(plan)
        m.train_steps(0, T)
        assert (m.get_debug('defer_stats', 4)[0] > 0) == bool(defer)
        outs.append((m.get_losses(0, T).copy(), m.get_param('Wy', (I, 48)).copy(), m.get_param('By', (I,)).copy()))
        m.close()
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            np.testing.assert_array_equal(a, b)


def test_deferral_is_off_where_it_does_not_apply():
    """Momentum (velocity rows), an L2 term and the generic optimizers keep the immediate update."""
    old = os.environ.get('G4R_DEFER')
    os.environ['G4R_DEFER'] = '1'
    try:
        for kw, want in ((dict(momentum=0.1), 0), (dict(lmbd=1e-4), 0), (dict(adapt='rmsprop', adapt_params=[0.9]), 0), (dict(), 1)):
            o, m = make_pair(500, 16, 32, store_rows=8, use_graph=1, loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(16,),
                             learning_rate=0.1, **kw)
            assert m.get_debug('defer_stats', 4)[2] == want, kw
            m.close()
    finally:
        if old is None:
            os.environ.pop('G4R_DEFER', None)
        else:
            os.environ['G4R_DEFER'] = old


def test_defer_updates_through_the_public_class():
    """GRU4Rec.defer_updates = True: one epoch ends with the same weights, bit for bit, as the default."""
    from gru4rec_amd import synth
    from gru4rec_amd.gru4rec import GRU4Rec
    data = synth.make_sessions(3000, n_items=800, seed=3)
    out = []
    for flag in (False, True):
        g = GRU4Rec(loss='bpr-max', final_act='elu-0.5', layers=[32], batch_size=32, n_sample=64, constrained_embedding=True, learning_rate=0.1,
                    bpreg=1.0, n_epochs=1)
        g.defer_updates = flag
        g.fit(data.copy(), sample_store=64 * 100)
        out.append((g.Wy.copy(), g.By.copy(), g.Wx[0].copy()))
        g.close()
    for a, b in zip(*out):
        np.testing.assert_array_equal(a, b)
