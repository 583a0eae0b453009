"""Checkpoint interchange with the reference (gru4rec.py:742-781).

tests/golden/checkpoint/ref_checkpoint.pickle was written by the REFERENCE's own `savemodel` (oracle/make_checkpoint_fixture.py runs
the reference source on the Theano stand-in); ref_checkpoint_pred.npz holds what its `predict_next_batch` returned."""
import os
import pickle

import numpy as np
import pytest

from gru4rec_amd.gru4rec import GRU4Rec

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'checkpoint')
PICKLE = os.path.join(GOLD, 'ref_checkpoint.pickle')
PRED = os.path.join(GOLD, 'ref_checkpoint_pred.npz')


def test_reference_pickle_loads_into_the_product_class():
    g = GRU4Rec.loadmodel(PICKLE)
    want = np.load(PRED, allow_pickle=False)
    assert isinstance(g, GRU4Rec)
    assert (g.loss, g.final_act, g.layers, g.embedding, g.constrained_embedding) == ('bpr-max', 'elu-0.5', [12], 8, False)
    np.testing.assert_array_equal(g.Wy, want['Wy'])
    np.testing.assert_array_equal(g.E, want['E'])
    np.testing.assert_array_equal(g.Wx[0], want['Wx0'])
    assert g.By.shape == (g.n_items, 1) and g.Wy.dtype == np.float32
    assert list(g.itemidmap.index.astype(str)) == list(want['itemids'])
    # the MI355X-side attributes the reference does not know get their defaults
    assert g.use_graph is True and g._model is None and g._loss_id is not None


def test_product_pickle_names_the_reference_class(tmp_path):
    """What the reference's loadmodel needs: module `gru4rec`, class `GRU4Rec`, NumPy arrays under the reference's
    attribute names, loss / activations as bound methods with the reference's method names."""
    g = GRU4Rec.loadmodel(PICKLE)
    f = str(tmp_path / 'm.pickle')
    g.savemodel(f)
    raw = open(f, 'rb').read()
    assert b'gru4rec_amd' not in raw and b'\x8c\x07gru4rec\x94\x8c\x07GRU4Rec' in raw
    h = GRU4Rec.loadmodel(f)
    st = h.__dict__
    for name in ('layers', 'loss', 'final_act', 'hidden_act', 'Wx', 'Wh', 'Wrz', 'Bh', 'H', 'Wy', 'By', 'E', 'itemidmap',
                 'n_items', 'loss_function', 'final_activation', 'hidden_activation', 'error_during_train'):
        assert name in st, name
    assert h.loss_function.__name__ == 'bpr_max' and h.hidden_activation.__name__ == 'tanh'
    assert type(h.final_activation.__self__).__name__ == 'Elu' and h.final_activation.__self__.alpha == 0.5
    assert '_model' not in pickle.loads(raw).__getstate__()
    np.testing.assert_array_equal(h.Wy, g.Wy)


def test_reference_reads_product_pickle_live(tmp_path):
    """Only where /root/reference exists: the reference's own loadmodel + predict_next_batch on a product-written pickle."""
    from oracle import ref_loader
    if not ref_loader.available():
        pytest.skip('/root/reference not present')
    import sys
    g = GRU4Rec.loadmodel(PICKLE)
    f = str(tmp_path / 'm.pickle')
    g.savemodel(f)
    saved = sys.modules.get('gru4rec')
    try:
        _, ref, _ = ref_loader.load()
        back = ref.GRU4Rec.loadmodel(f)
        want = np.load(PRED, allow_pickle=False)
        ids = np.array(list(back.itemidmap.index))
        p1 = back.predict_next_batch(np.array([1, 2, 3, 4]), ids[[0, 3, 5, 7]], None, batch=4)
        np.testing.assert_array_equal(p1.values.astype(np.float32), want['pred1'])
    finally:
        ref_loader.unload()
        if saved is not None:
            sys.modules['gru4rec'] = saved


@pytest.mark.gpu
def test_reference_checkpoint_predicts_on_the_gpu():
    g = GRU4Rec.loadmodel(PICKLE)
    want = np.load(PRED, allow_pickle=False)
    ids = np.array(list(g.itemidmap.index))
    p1 = g.predict_next_batch(np.array([1, 2, 3, 4]), ids[[0, 3, 5, 7]], None, batch=4)
    p2 = g.predict_next_batch(np.array([1, 2, 9, 4]), ids[[2, 3, 1, 6]], None, batch=4)
    np.testing.assert_allclose(p1.values, want['pred1'], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(p2.values, want['pred2'], rtol=2e-4, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('kw', [dict(momentum=0.2, dropout_p_hidden=0.2, dropout_p_embed=0.1),
                                dict(adapt='adam', adapt_params=[0.9, 0.999], learning_rate=0.01, constrained_embedding=False, embedding=16),
                                dict(constrained_embedding=False, layers=[24]),
                                dict(train_random_order=True, momentum=0.1)],
                         ids=['adagrad_momentum_dropout', 'adam_separate_embedding', 'onehot', 'random_session_order'])
def test_resume_continues_bit_identically(tmp_path, kw):
    """SURVEY 8f rank 2, second half: 2 epochs == 1 epoch + savemodel(optimizer_state=True) + loadmodel + fit(resume=True).
    The sample store wraps around inside the epochs (its refill counter is part of the saved state), dropout masks are keyed by
    the global step."""
    from gru4rec_amd import synth
    data = synth.make_sessions(1500, n_items=300, seed=4)
    P = dict(loss='bpr-max', final_act='elu-0.5', layers=[32], batch_size=32, n_sample=64, constrained_embedding=True, learning_rate=0.1)
    P.update(kw)
    a = GRU4Rec(n_epochs=2, **P)
    a.fit(data.copy(), sample_store=64 * 50)
    b = GRU4Rec(n_epochs=1, **P)
    b.fit(data.copy(), sample_store=64 * 50)
    f = str(tmp_path / 'ckpt.pickle')
    b.savemodel(f, optimizer_state=True)
    b.close()
    c = GRU4Rec.loadmodel(f)
    assert c.epochs_done == 1 and c.optimizer_state['global_step'] == len(a.step_costs[0])
    c.n_epochs = 2
    c.fit(data.copy(), sample_store=64 * 50, resume=True)
    np.testing.assert_array_equal(c.step_costs[0], a.step_costs[1])
    np.testing.assert_array_equal(c.Wy, a.Wy)
    np.testing.assert_array_equal(c.By, a.By)
    for i in range(len(a.layers)):
        np.testing.assert_array_equal(c.Wx[i], a.Wx[i])
        np.testing.assert_array_equal(c.Wh[i], a.Wh[i])
    assert c.loss_history == a.loss_history
    if kw.get('train_random_order'):
        # the session order of the resumed epoch is the uninterrupted run's second permutation (NumPy's global stream travels in
        # the checkpoint), not a fresh draw
        assert not np.array_equal(a.step_costs[0], a.step_costs[1][:len(a.step_costs[0])])
    # a checkpoint without optimizer state refuses to resume instead of silently restarting the accumulators
    b2 = GRU4Rec.loadmodel(f)
    b2.optimizer_state = None
    with pytest.raises(ValueError):
        b2.fit(data.copy(), sample_store=64 * 50, resume=True)
    # ... and so does one whose optimizer state belongs to another configuration (the velocities would be dropped silently)
    b3 = GRU4Rec.loadmodel(f)
    b3.n_epochs = 2
    b3.momentum = 0.0 if b3.momentum > 0 else 0.3
    with pytest.raises(ValueError, match='momentum'):
        b3.fit(data.copy(), sample_store=64 * 50, resume=True)
    # the host sampler cannot resume, and says so before anything is built
    b4 = GRU4Rec.loadmodel(f)
    b4.n_epochs = 2
    with pytest.raises(NotImplementedError):
        b4.fit(data.copy(), sample_store=64 * 50, store_type='cpu', resume=True)
    assert b4._model is None


def test_checkpoint_with_optimizer_state_keeps_the_reference_layout(tmp_path):
    """The extra attributes ride along; everything the reference's loadmodel reads is unchanged."""
    g = GRU4Rec.loadmodel(PICKLE)
    g.optimizer_state = {'arrays': {('acc_Wy', 0): np.ones((g.n_items, 12), dtype=np.float32)}, 'global_step': 7, 'refills': 1}
    g.epochs_done = 3
    f = str(tmp_path / 'm.pickle')
    with open(f, 'wb') as fh:
        pickle.dump(g, fh)
    h = GRU4Rec.loadmodel(f)
    assert h.epochs_done == 3 and h.optimizer_state['global_step'] == 7
    np.testing.assert_array_equal(h.Wy, g.Wy)
    from oracle import ref_loader
    if ref_loader.available():
        import sys
        saved = sys.modules.get('gru4rec')
        try:
            _, ref, _ = ref_loader.load()
            back = ref.GRU4Rec.loadmodel(f)      # the reference's own loader accepts the file
            assert back.epochs_done == 3
        finally:
            ref_loader.unload()
            if saved is not None:
                sys.modules['gru4rec'] = saved
