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


@pytest.mark.parametrize('mode', ['standard', 'conservative', 'median', 'tiebreaking'])
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


@pytest.mark.parametrize('mode', ['standard', 'conservative', 'median', 'tiebreaking'])
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


def test_native_loaded_table_trains_and_evaluates_like_the_pandas_table(tmp_path):
    """fit + evaluate_gpu on tables from eventio.read_events (categorical item column, integer fast paths) and from
    pandas.read_csv (str column, the reference's expressions): same item map, same per-step costs, same metrics."""
    import pandas as pd
    from gru4rec_amd import eventio
    data = synth.make_sessions(1500, n_items=300, seed=21)
    train, test = synth.train_test_split(data)
    paths = {}
    for name, frame in (('train', train), ('test', test)):
        paths[name] = str(tmp_path / (name + '.tsv'))
        frame[['SessionId', 'ItemId', 'Time']].to_csv(paths[name], sep='\t', index=False)
    out = []
    for engine in ('native', 'pandas'):
        tr = eventio.read_events(paths['train'], engine=engine)
        te = eventio.read_events(paths['test'], engine=engine)
        assert eventio.is_categorical(tr['ItemId']) == (engine == 'native')
        gru = GRU4Rec(loss='bpr-max', final_act='elu-0.5', layers=[32], batch_size=32, n_sample=64, constrained_embedding=True,
                      n_epochs=2, learning_rate=0.1, dropout_p_embed=0.1)
        gru.fit(tr, sample_store=64 * 50)
        res = evaluation.evaluate_gpu(gru, te, batch_size=16, cut_off=[1, 5, 20])
        out.append((gru.itemidmap, np.concatenate(gru.step_costs), gru.Wy.copy(), res, tr['ItemIdx'].values.copy()))
    (map_n, cost_n, wy_n, res_n, idx_n), (map_p, cost_p, wy_p, res_p, idx_p) = out
    assert map_n.index.tolist() == map_p.index.tolist() and np.array_equal(map_n.values, map_p.values)
    assert np.array_equal(idx_n, idx_p)
    assert np.array_equal(cost_n, cost_p) and np.array_equal(wy_n, wy_p)
    assert res_n == res_p


def test_paropt_in_process_trials_on_the_device(tmp_path, capsys):
    """paropt.py end to end: two sampled points trained and scored in this process, the winner re-evaluated."""
    import json
    import paropt
    data = synth.make_sessions(4000, n_items=200, seed=5)
    train, test = synth.train_test_split(data, test_frac=0.2)
    for name, frame in (('train', train), ('test', test)):
        frame[['SessionId', 'ItemId', 'Time']].to_csv(str(tmp_path / (name + '.tsv')), sep='\t', index=False)
    (tmp_path / 'space.json').write_text(json.dumps({'name': 'learning_rate', 'dtype': 'float', 'values': [0.05, 0.2], 'step': 0.05}) + '\n' +
                                         json.dumps({'name': 'layers', 'dtype': 'int', 'values': [16, 48], 'step': 16}) + '\n')
    best_value, best_point = paropt.main([str(tmp_path / 'train.tsv'), str(tmp_path / 'test.tsv'), '-fp',
                                          'loss=bpr-max,final_act=elu-0.5,batch_size=32,n_sample=64,n_epochs=1,constrained_embedding=True',
                                          '-opf', str(tmp_path / 'space.json'), '-nt', '2', '-fm', '5', '20', '--sampler', 'random'])
    out = capsys.readouterr().out
    assert out.count('PRIMARY METRIC: ') == 2 and out.count('Epoch1 --> loss:') == 3
    assert 0.0 < best_value <= 1.0 and set(best_point) == {'learning_rate', 'layers'}
    assert 'Recall@5: ' in out and 'Recall@20: ' in out
