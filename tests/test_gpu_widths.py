"""Layer / embedding widths that are not multiples of 4 (the reference has no such restriction; the device layout wants 16-byte
rows, so the public class pads with zero units on the way to the device and strips them on the way back).  A short training run
through the public class against the oracle at the TRUE widths: per-step costs, final weights and the optimizer state that a
checkpoint would hold."""
import numpy as np
import pytest

from gru4rec_amd import synth
from gru4rec_amd.gru4rec import GRU4Rec
from oracle.driver import oracle_fit

pytestmark = pytest.mark.gpu

CASES = {
    'constrained_50': dict(layers=[50], constrained_embedding=True, loss='bpr-max', final_act='linear', momentum=0.1, dropout_p_hidden=0.2),
    'two_layers_30_22': dict(layers=[30, 22], constrained_embedding=True, loss='cross-entropy', final_act='softmax', logq=1.0,
                             dropout_p_embed=0.2),
    'embedding_25_layer_18': dict(layers=[18], constrained_embedding=False, embedding=25, loss='top1-max', final_act='tanh'),
    'onehot_10': dict(layers=[10], constrained_embedding=False, embedding=0, loss='bpr-max', final_act='linear'),
    'adam_37': dict(layers=[37], constrained_embedding=True, loss='bpr-max', final_act='linear', adapt='adam', adapt_params=[0.9, 0.999],
                    learning_rate=0.01),
}


@pytest.mark.parametrize('name', sorted(CASES))
def test_unaligned_widths_train_like_the_oracle(name):
    kw = dict(batch_size=16, n_sample=48, learning_rate=0.1, n_epochs=1, sample_alpha=0.5)
    kw.update(CASES[name])
    data = synth.make_sessions(400, n_items=120, seed=7)
    gru = GRU4Rec(**kw)
    gru.fit(data.copy(), sample_store=48 * 40)
    p = dict(kw)
    p['layers'] = tuple(p['layers'])
    p['adapt_params'] = tuple(p.get('adapt_params', ()))
    run = oracle_fit(data.copy(), p, 48 * 40, seed=gru.seed)
    got = np.concatenate(gru.step_costs)
    assert len(got) == len(run.costs) > 40
    np.testing.assert_allclose(got, run.costs, rtol=1e-3, atol=1e-5)
    o = run.model
    for i, D in enumerate(kw['layers']):
        assert gru.Wx[i].shape == o.Wx[i].shape and gru.Wh[i].shape == (D, D) and gru.Bh[i].shape == (3 * D,)
        np.testing.assert_allclose(gru.Wx[i], o.Wx[i], rtol=5e-3, atol=2e-4)
        np.testing.assert_allclose(gru.Wh[i], o.Wh[i], rtol=5e-3, atol=2e-4)
        np.testing.assert_allclose(gru.Wrz[i], o.Wrz[i], rtol=5e-3, atol=2e-4)
        np.testing.assert_allclose(gru.Bh[i], o.Bh[i], rtol=5e-3, atol=2e-4)
    assert gru.Wy.shape == o.Wy.shape
    np.testing.assert_allclose(gru.Wy, o.Wy, rtol=5e-3, atol=2e-4)
    np.testing.assert_allclose(gru.By.reshape(-1), o.By, rtol=5e-3, atol=2e-4)
    if kw.get('embedding'):
        np.testing.assert_allclose(gru.E, o.E, rtol=5e-3, atol=2e-4)
    st = gru._download_optimizer_state()['arrays']
    np.testing.assert_allclose(st[('acc_Wy', 0)], o.acc['Wy'], rtol=1e-2, atol=1e-7)
    assert st[('acc_Wh', 0)].shape == o.acc['Wh'][0].shape
    # prediction goes through the same padding
    ids = np.array(list(gru.itemidmap.index))[:4]
    pr = gru.predict_next_batch(np.arange(4), ids, None, batch=4)
    assert pr.shape == (gru.n_items, 4) and np.isfinite(pr.values).all()
    gru.close()


WIDE = {
    # rows of more than 512 floats: four quads per lane in the sparse update, dense gradients and sparse update as two launches
    'constrained_640': dict(layers=[640], constrained_embedding=True, loss='bpr-max', final_act='elu-0.5', bpreg=0.5, learning_rate=0.01),      # (0.05 diverges, in the oracle too)
    'embedding_600_layer_64': dict(layers=[64], constrained_embedding=False, embedding=600, loss='cross-entropy', final_act='softmax', logq=1.0),
    'onehot_200': dict(layers=[200], constrained_embedding=False, embedding=0, loss='top1-max', final_act='tanh'),      # Wx[0] rows of 600
    'adadelta_1000': dict(layers=[1000], constrained_embedding=True, loss='bpr-max', final_act='linear', adapt='adadelta', adapt_params=[0.95],
                          learning_rate=0.5),
}


@pytest.mark.parametrize('name', sorted(WIDE))
def test_rows_wider_than_512_floats(name):
    kw = dict(batch_size=16, n_sample=48, learning_rate=0.05, n_epochs=1, sample_alpha=0.5)
    kw.update(WIDE[name])
    data = synth.make_sessions(160, n_items=90, seed=11)
    gru = GRU4Rec(**kw)
    gru.fit(data.copy(), sample_store=48 * 40)
    p = dict(kw)
    p['layers'] = tuple(p['layers'])
    p['adapt_params'] = tuple(p.get('adapt_params', ()))
    run = oracle_fit(data.copy(), p, 48 * 40, seed=gru.seed)
    got = np.concatenate(gru.step_costs)
    assert len(got) == len(run.costs) > 15
    np.testing.assert_allclose(got, run.costs, rtol=2e-3, atol=1e-5)
    o = run.model
    np.testing.assert_allclose(gru.Wy, o.Wy, rtol=5e-3, atol=2e-4)
    np.testing.assert_allclose(gru.Wx[0], o.Wx[0], rtol=5e-3, atol=2e-4)
    np.testing.assert_allclose(gru.Wh[0], o.Wh[0], rtol=5e-3, atol=2e-4)
    if kw.get('embedding'):
        np.testing.assert_allclose(gru.E, o.E, rtol=5e-3, atol=2e-4)
    ids = np.array(list(gru.itemidmap.index))[:4]      # the prediction path at the same widths
    pr = gru.predict_next_batch(np.arange(4), ids, None, batch=4)
    assert pr.shape == (gru.n_items, 4) and np.isfinite(pr.values).all()
    gru.close()
