"""Oracle drivers (TEST INFRASTRUCTURE): the data preparation and loops of the reference's `GRU4Rec.fit`
(gru4rec.py:532-545,585-661), `predict_next_batch` (:665-728) and `evaluation.evaluate_gpu`
(evaluation.py:77-147) around `oracle.model.OracleGRU4Rec`."""
import numpy as np
import pandas as pd

from .model import OracleGRU4Rec, ranks_from_scores
from .scheduler import eval_schedule, fit_schedule


class OracleRun:
    pass


def oracle_fit(data, params, sample_store, seed=12345, dtype=np.float32, session_key='SessionId', item_key='ItemId',
               time_key='Time', max_steps=None, store_type='gpu'):
    """Returns an OracleRun with .model, .costs (per step, all epochs), .M (per step), .itemidmap."""
    p = dict(params)
    n_epochs = p.pop('n_epochs', 10)
    time_sort = p.pop('time_sort', True)
    random_order = p.pop('train_random_order', False)
    data = data.copy()
    itemids = data[item_key].unique()
    n_items = len(itemids)
    itemidmap = pd.Series(data=np.arange(n_items), index=itemids, name='ItemIdx')
    data['ItemIdx'] = itemidmap[data[item_key].values].values
    data.sort_values([session_key, time_key], inplace=True, kind='stable')
    sizes = data.groupby(session_key).size().values
    offsets = np.zeros(len(sizes) + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(sizes)
    model = OracleGRU4Rec(n_items=n_items, dtype=dtype, seed=seed, **p)
    support = data.groupby(item_key).size()[itemidmap.index.values].values
    model.set_popularity(support)
    model.make_sample_store(sample_store, store_type)
    if time_sort:
        order = np.argsort(data.groupby(session_key)[time_key].min().values)
    else:
        order = np.arange(len(offsets) - 1)
    items = data.ItemIdx.values
    run = OracleRun()
    run.model, run.itemidmap, run.costs, run.M, run.epoch_loss = model, itemidmap, [], [], []
    for epoch in range(n_epochs):
        for i in range(len(model.layers)):
            model.H[i] = np.zeros((model.batch_size, model.layers[i]), dtype=dtype)
        c, cc = [], []
        if random_order:
            order = np.random.permutation(len(offsets) - 1)      # gru4rec.py:592-593
        for ev in fit_schedule(offsets, order, items, model.batch_size, model.n_sample):
            if ev[0] == 'step':
                c.append(model.train_step(ev[1], ev[2], ev[3], ev[4]))
                cc.append(ev[3])
                if max_steps is not None and len(run.costs) + len(c) >= max_steps:
                    break
            else:
                valid = ev[1]
                for i in range(len(model.layers)):       # gru4rec.py:647-651
                    H = np.zeros_like(model.H[i])
                    keep = model.H[i][:len(valid)][valid]
                    H[:len(keep)] = keep
                    model.H[i] = H
        run.costs += c
        run.M += cc
        c, cc = np.array(c), np.array(cc)
        run.epoch_loss.append(float(np.sum(c * cc) / np.sum(cc)))
    run.costs = np.array(run.costs, dtype=np.float32)
    run.M = np.array(run.M)
    return run


class OraclePredictor:
    """predict_next_batch semantics (gru4rec.py:691-728): hidden rows are zeroed when the session id changes."""

    def __init__(self, model, itemidmap, batch):
        self.model, self.itemidmap, self.batch = model, itemidmap, batch
        self.H = [np.zeros((batch, D), dtype=model.dtype) for D in model.layers]
        self.current = np.ones(batch) * -1

    def predict_next_batch(self, session_ids, input_item_ids, predict_for_item_ids=None):
        session_ids = np.asarray(session_ids)
        changed = session_ids != self.current
        for h in self.H:
            h[changed] = 0
        self.current = session_ids.copy()
        in_idx = self.itemidmap[input_item_ids].values
        sel = None if predict_for_item_ids is None else self.itemidmap[predict_for_item_ids].values
        yhat, self.H = self.model.predict_step(self.H, in_idx, sel)
        return yhat.T


def oracle_evaluate(model, itemidmap, test_data, cut_off=(20,), batch_size=100, mode='standard',
                    session_key='SessionId', item_key='ItemId', time_key='Time'):
    """evaluation.py:77-147 (all-items case).  Returns (recall list, mrr list)."""
    lookup = pd.DataFrame({'ItemIdx': itemidmap.values, item_key: itemidmap.index})
    test = pd.merge(test_data, lookup, on=item_key, how='inner')
    test.sort_values([session_key, time_key, item_key], inplace=True)
    titems = test.ItemIdx.values
    sizes = test.groupby(session_key).size().values
    offs = np.zeros(len(sizes) + 1, dtype=np.int64)
    offs[1:] = np.cumsum(sizes)
    H = [np.zeros((batch_size, D), dtype=model.dtype) for D in model.layers]
    rec = np.zeros(len(cut_off))
    mrr = np.zeros(len(cut_off))
    n = 0
    step = 0
    for ev in eval_schedule(offs, titems, batch_size):
        if ev[0] == 'step':
            _, cur_in, cur_out, M = ev
            yhat, H = model.predict_step(H, cur_in)
            ranks = ranks_from_scores(yhat, cur_out, mode, tie=(model.seed, step))
            step += 1
            for j, c in enumerate(cut_off):
                hit = ranks <= c
                rec[j] += hit.sum()
                mrr[j] += (hit / ranks).sum()
            n += M
        else:
            _, zero, valid = ev
            for i in range(len(H)):
                H[i][zero] = 0
                H[i] = H[i][valid]
    return (rec / n).tolist(), (mrr / n).tolist()
