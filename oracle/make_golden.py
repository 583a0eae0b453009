#!/usr/bin/env python
"""Generate golden vectors by executing the REFERENCE's own source (/root/reference/gru4rec.py,
gpu_ops.py, datatools.py, evaluation.py) on top of the Theano stand-in in oracle/theano_shim.

    python oracle/make_golden.py            # writes tests/golden/*.npz  (run in the build container only)

What is pinned: for each scenario the reference `GRU4Rec.fit` is run on a small deterministic click-stream;
we record the per-mini-batch cost returned by its `train_function` (gru4rec.py:623), the final parameters,
`predict_next_batch` scores and `evaluation.evaluate_gpu` Recall/MRR.  Randomness (negative-sample uniforms,
dropout masks) is routed through the shim's RNG hook to the Philox streams of oracle/philox.py, so the
oracle and the HIP path can consume the identical draws.

TEST INFRASTRUCTURE: nothing in the product imports this; /root/reference is only needed to (re)generate.
"""
import os
import sys

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = '/root/reference'
sys.path.insert(0, os.path.join(HERE, 'theano_shim'))
sys.path.insert(1, ROOT)
sys.path.append(REF)

import theano  # noqa: E402  (the shim)
from oracle import philox  # noqa: E402

SEED = 12345

SCENARIOS = {
    'bprmax_constrained': dict(loss='bpr-max', final_act='elu-0.5', hidden_act='tanh', layers=[12], n_epochs=2,
                               batch_size=8, dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.2,
                               momentum=0.1, n_sample=16, sample_alpha=0.5, bpreg=0.5, constrained_embedding=True),
    'xe_logq_dropout': dict(loss='cross-entropy', final_act='softmax', hidden_act='tanh', layers=[12], n_epochs=2,
                            batch_size=8, dropout_p_embed=0.0, dropout_p_hidden=0.4, learning_rate=0.2,
                            momentum=0.2, n_sample=16, sample_alpha=0.5, bpreg=0.0, logq=1.0,
                            constrained_embedding=True),
    'top1max_2layer_embdrop': dict(loss='top1-max', final_act='tanh', hidden_act='tanh', layers=[8, 12], n_epochs=2,
                                   batch_size=8, dropout_p_embed=0.25, dropout_p_hidden=0.1, learning_rate=0.1,
                                   momentum=0.0, n_sample=16, sample_alpha=0.75, constrained_embedding=True),
    'xe_separate_embedding': dict(loss='cross-entropy', final_act='softmax', hidden_act='relu', layers=[12],
                                  n_epochs=1, batch_size=6, dropout_p_embed=0.0, dropout_p_hidden=0.0,
                                  learning_rate=0.1, momentum=0.0, n_sample=0, embedding=8,
                                  constrained_embedding=False),
    'onehot_default': dict(loss='bpr-max', final_act='elu-0.5', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                           dropout_p_embed=0.0, dropout_p_hidden=0.2, learning_rate=0.1, momentum=0.1, n_sample=16,
                           sample_alpha=0.75, bpreg=1.0, constrained_embedding=False, embedding=0),
    'rmsprop_mom': dict(loss='bpr-max', final_act='elu-0.5', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                        dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.01, momentum=0.2, n_sample=16,
                        sample_alpha=0.5, bpreg=1.0, constrained_embedding=True, adapt='rmsprop', adapt_params=[0.9]),
    'adadelta_xe': dict(loss='cross-entropy', final_act='softmax', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                        dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=1.0, momentum=0.0, n_sample=16,
                        sample_alpha=0.5, constrained_embedding=True, adapt='adadelta', adapt_params=[0.95]),
    'adam_sep_embed': dict(loss='bpr-max', final_act='elu-0.5', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                           dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.01, momentum=0.0, n_sample=16,
                           sample_alpha=0.5, bpreg=1.0, constrained_embedding=False, embedding=8, adapt='adam',
                           adapt_params=[0.9, 0.999]),
    'sgd_gradcap': dict(loss='top1-max', final_act='tanh', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                        dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.05, momentum=0.1, n_sample=16,
                        sample_alpha=0.5, constrained_embedding=True, adapt=None, grad_cap=0.05),
    'adagrad_gradcap_lmbd': dict(loss='bpr-max', final_act='elu-0.5', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                                 dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.1, momentum=0.0, n_sample=16,
                                 sample_alpha=0.5, bpreg=1.0, constrained_embedding=True, grad_cap=0.02, lmbd=0.01),
    'bpr_linear': dict(loss='bpr', final_act='linear', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                       dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.1, momentum=0.0, n_sample=16,
                       sample_alpha=0.5, constrained_embedding=True),
    'top1_tanh': dict(loss='top1', final_act='tanh', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                      dropout_p_embed=0.0, dropout_p_hidden=0.2, learning_rate=0.1, momentum=0.1, n_sample=16,
                      sample_alpha=0.75, constrained_embedding=True),
    'xelogit_smoothing': dict(loss='xe_logit', final_act='softmax_logit', hidden_act='tanh', layers=[12], n_epochs=2,
                              batch_size=8, dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.2,
                              momentum=0.0, n_sample=16, sample_alpha=0.5, smoothing=0.1, logq=1.0,
                              constrained_embedding=True),
    # store_type='cpu' (README.md:470-475 route): the reference samples on the host with NumPy's global stream, right behind its
    # weight initialisation -- no random numbers are substituted in these runs (no dropout): negatives, session order and weights
    # are the reference's own draws
    'cpu_store_bprmax': dict(loss='bpr-max', final_act='elu-0.5', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                             dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.2, momentum=0.1, n_sample=16,
                             sample_alpha=0.5, bpreg=0.5, constrained_embedding=True, store_type='cpu'),
    'cpu_store_uniform_random_order': dict(loss='top1-max', final_act='tanh', hidden_act='tanh', layers=[12], n_epochs=2, batch_size=8,
                                           dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.1, momentum=0.0, n_sample=16,
                                           sample_alpha=0.0, constrained_embedding=True, train_random_order=True, store_type='cpu'),
    'cpu_no_store_per_step': dict(loss='cross-entropy', final_act='softmax', hidden_act='tanh', layers=[12], n_epochs=1, batch_size=8,
                                  dropout_p_embed=0.0, dropout_p_hidden=0.0, learning_rate=0.1, momentum=0.0, n_sample=16,
                                  sample_alpha=0.75, logq=1.0, constrained_embedding=True, store_type='cpu', sample_store_rows=1),
}
SAMPLE_STORE_ROWS = 9        # generate_length: small, so that the store is refilled several times


def make_data(seed, n_sessions=70, n_items=40):
    rng = np.random.RandomState(seed)
    rows = []
    t = 1000
    for s in range(n_sessions):
        ln = rng.randint(1, 8)
        cur = rng.randint(0, n_items)
        for _ in range(ln):
            rows.append((s + 1, 'i%03d' % cur, t))
            t += rng.randint(1, 30)
            cur = (cur * 7 + rng.randint(0, 3)) % n_items if rng.rand() < 0.7 else rng.randint(0, n_items)
    return pd.DataFrame(rows, columns=['SessionId', 'ItemId', 'Time'])


def install_rng_hook(params, state):
    has_store = params.get('n_sample', 0) > 0

    def hook(kind, rank, call_count, shape, attrs):
        if kind == 'uniform':
            n = int(np.prod(shape))
            u = philox.uniform_block(n, SEED, state['refills'], 0, philox.STREAM_SAMPLE)
            state['refills'] += 1
            return u.reshape(shape)
        # dropout sites in creation order (gru4rec.py:443,477): [embedding dropout] then one per GRU layer
        site = rank - (1 if has_store else 0)
        sites = []
        if params.get('dropout_p_embed', 0) > 0:
            sites.append(philox.STREAM_DROP_EMBED)
        if params.get('dropout_p_hidden', 0) > 0:
            sites += [philox.STREAM_DROP_HIDDEN + i for i in range(len(params['layers']))]
        stream = sites[site]
        retain = float(attrs['p'])
        m = philox.dropout_mask(shape[0], shape[1], retain, SEED, call_count, stream)
        return (m * np.float32(retain)).round().astype(np.float32)      # 0/1: the reference divides by retain itself
    theano.RNG_HOOK[0] = hook


def run_scenario(name, params):
    from oracle import ref_loader
    theano._rng_nodes.clear()
    theano.CALL_LOG.clear()
    _, ref_gru4rec, ref_eval = ref_loader.load()      # the reference's own modules, by explicit path
    data = make_data(7)
    train = data[data.SessionId <= 55].copy()
    test = data[data.SessionId > 55].copy()
    test = test[np.isin(test.ItemId, train.ItemId)]
    state = {'refills': 0}
    install_rng_hook(params, state)
    params = dict(params)
    store_type = params.pop('store_type', 'gpu')
    rows = params.pop('sample_store_rows', SAMPLE_STORE_ROWS)
    gru = ref_gru4rec.GRU4Rec(**params)
    store = rows * params['n_sample'] if params['n_sample'] else 0
    gru.fit(train.copy(), sample_store=store, store_type=store_type)
    costs = np.array([np.asarray(o).reshape(()) for (fid, o) in theano.CALL_LOG if o is not None], dtype=np.float32)
    out = dict(costs=costs, n_items=gru.n_items, itemids=np.array(list(gru.itemidmap.index)))
    for i in range(len(params['layers'])):
        out['Wx%d' % i] = gru.Wx[i].get_value()
        out['Wh%d' % i] = gru.Wh[i].get_value()
        out['Wrz%d' % i] = gru.Wrz[i].get_value()
        out['Bh%d' % i] = gru.Bh[i].get_value()
    out['Wy'] = gru.Wy.get_value()
    out['By'] = gru.By.get_value()
    if params.get('embedding'):
        out['E'] = gru.E.get_value()
    # prediction (gru4rec.py:665-728): two consecutive batches, second one with a session change
    theano.CALL_LOG.clear()
    ids = np.array(list(gru.itemidmap.index))
    sess = np.array([1, 2, 3, 4])
    p1 = gru.predict_next_batch(sess, ids[[0, 3, 5, 7]], None, batch=4)
    sess2 = np.array([1, 2, 9, 4])
    p2 = gru.predict_next_batch(sess2, ids[[2, 3, 1, 6]], None, batch=4)
    out['pred1'] = p1.values.astype(np.float32)
    out['pred2'] = p2.values.astype(np.float32)
    out['pred_in1'] = ids[[0, 3, 5, 7]]
    out['pred_in2'] = ids[[2, 3, 1, 6]]
    # evaluation (evaluation.py:15-147), all-items mode, three tie modes
    for mode in ('standard', 'conservative', 'median'):
        rec, mrr = ref_eval.evaluate_gpu(gru, test.copy(), cut_off=[1, 5, 20], batch_size=5, mode=mode)
        out['recall_' + mode] = np.array(rec, dtype=np.float64)
        out['mrr_' + mode] = np.array(mrr, dtype=np.float64)
    for c in ('SessionId', 'ItemId', 'Time'):
        out['train_' + c] = train[c].values
        out['test_' + c] = test[c].values
    out['params'] = np.array(repr(params))
    out['sample_store'] = store
    out['store_type'] = np.array(store_type)
    out['seed'] = SEED
    out = {k: (np.asarray(v).astype(str) if np.asarray(v).dtype == object else v) for k, v in out.items()}
    path = os.path.join(ROOT, 'tests', 'golden', name + '.npz')
    np.savez_compressed(path, **out)
    print('%-28s steps %4d  cost[0] %.6f cost[-1] %.6f  recall@20 %.4f mrr@20 %.4f -> %s' % (
        name, len(costs), costs[0], costs[-1], out['recall_standard'][2], out['mrr_standard'][2], path))


if __name__ == '__main__':
    os.makedirs(os.path.join(ROOT, 'tests', 'golden'), exist_ok=True)
    which = sys.argv[1:] or sorted(SCENARIOS)
    cwd = os.getcwd()
    for n in which:
        run_scenario(n, SCENARIOS[n])
        os.chdir(cwd)
