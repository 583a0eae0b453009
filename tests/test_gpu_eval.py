"""Device-resident evaluation (g4r_evaluate, one call per test set) against the host-driven step-by-step loop: same
Recall@N / MRR@N to the last bit of the hit counts, all tie modes, all-items and item-subset candidates, shrinking tail."""
import numpy as np
import pytest

from gru4rec_amd import evaluation, synth
from gru4rec_amd.gru4rec import GRU4Rec

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def trained():
    data = synth.make_sessions(2500, n_items=400, seed=9)
    train, test = synth.train_test_split(data)
    gru = GRU4Rec(loss='cross-entropy', final_act='softmax', layers=[32], batch_size=32, n_sample=64, constrained_embedding=True,
                  n_epochs=2, learning_rate=0.1)
    gru.fit(train, sample_store=64 * 50)
    return gru, test


@pytest.fixture(scope='module', params=['elu-0.5', 'relu'])
def trained_elementwise(request):
    """Element-wise final activations take the streaming path (no score matrix); relu produces many exact ties."""
    data = synth.make_sessions(9000, n_items=400, seed=11)
    train, test = synth.train_test_split(data)
    gru = GRU4Rec(loss='bpr-max', final_act=request.param, layers=[36], batch_size=32, n_sample=64, constrained_embedding=True,
                  n_epochs=2, learning_rate=0.1)
    gru.fit(train, sample_store=64 * 50)
    return gru, test


@pytest.mark.parametrize('mode', ['standard', 'conservative', 'median'])
@pytest.mark.parametrize('batch', [5, 130])
def test_streaming_ranks_equal_materialised_ranks(trained_elementwise, mode, batch):
    gru, test = trained_elementwise
    a = evaluation.evaluate_gpu(gru, test.copy(), cut_off=[1, 5, 20], batch_size=batch, mode=mode)
    b = evaluation.evaluate_gpu_stepwise(gru, test.copy(), cut_off=[1, 5, 20], batch_size=batch, mode=mode)
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=1e-9)
    items = np.array(list(gru.itemidmap.index))[1::4]
    a = evaluation.evaluate_gpu(gru, test.copy(), items=items, cut_off=[2, 10], batch_size=batch, mode=mode)
    b = evaluation.evaluate_gpu_stepwise(gru, test.copy(), items=items, cut_off=[2, 10], batch_size=batch, mode=mode)
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize('mode', ['standard', 'conservative', 'median'])
@pytest.mark.parametrize('batch', [7, 50])
def test_one_call_equals_stepwise_all_items(trained, mode, batch):
    gru, test = trained
    a = evaluation.evaluate_gpu(gru, test.copy(), cut_off=[1, 5, 20], batch_size=batch, mode=mode)
    b = evaluation.evaluate_gpu_stepwise(gru, test.copy(), cut_off=[1, 5, 20], batch_size=batch, mode=mode)
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=1e-9)      # the host loop sums float32 reciprocal ranks, the device doubles
    assert 0 < a[0][-1] <= 1


def test_one_call_equals_stepwise_item_subset(trained):
    gru, test = trained
    items = np.array(list(gru.itemidmap.index))[::3]
    a = evaluation.evaluate_gpu(gru, test.copy(), items=items, cut_off=[3, 10], batch_size=20, mode='standard')
    b = evaluation.evaluate_gpu_stepwise(gru, test.copy(), items=items, cut_off=[3, 10], batch_size=20, mode='standard')
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-12)
    np.testing.assert_allclose(a[1], b[1], rtol=1e-6, atol=1e-9)      # the host loop sums float32 reciprocal ranks, the device doubles


def test_scalar_cut_off_and_errors(trained):
    gru, test = trained
    r, m = evaluation.evaluate_gpu(gru, test.copy(), cut_off=20, batch_size=16)
    assert len(r) == 1 and len(m) == 1
    with pytest.raises(IndexError):
        evaluation.evaluate_gpu(gru, test.iloc[:30].copy(), cut_off=[20], batch_size=512)
