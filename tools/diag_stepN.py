"""Diagnostic: at step N (state synced from the device), compare d cost / d s element-wise with the oracle."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gru4rec_amd import synth, _native
from gru4rec_amd.gru4rec import GRU4Rec
from oracle.model import OracleGRU4Rec
N_STEP = int(sys.argv[1])
data = synth.make_sessions(24000, n_items=2500, seed=17)
train, test = synth.train_test_split(data, test_frac=0.1)
P = dict(loss='bpr-max', final_act='elu-0.5', layers=[100], batch_size=128, n_sample=2048, constrained_embedding=True,
         learning_rate=0.1, bpreg=1.0, momentum=0.0, sample_alpha=0.75, n_epochs=1)
gru = GRU4Rec(**P); gru.use_graph = False
gru.prepare(train.copy(), sample_store=2048 * 640)
m = gru._model; plan = gru._epoch_plan(); m.reset_hidden()
I, D, B, ns = gru.n_items, 100, 128, 2048
o = OracleGRU4Rec(n_items=I, layers=(100,), batch_size=128, loss='bpr-max', final_act='elu-0.5', n_sample=2048,
                  constrained_embedding=True, learning_rate=0.1, bpreg=1.0, sample_alpha=0.75, dtype=np.float32, seed=gru.seed)
sup = np.bincount(train.assign(ItemIdx=gru.itemidmap[train.ItemId.values].values).ItemIdx.values, minlength=I)
o.set_popularity(sup); o.make_sample_store(2048 * 640)
if N_STEP:
    m.train_steps(0, N_STEP)
o.Wx[0] = m.get_param('Wx', (D, 3 * D), 0); o.Wh[0] = m.get_param('Wh', (D, D), 0); o.Wrz[0] = m.get_param('Wrz', (D, 2 * D), 0)
o.Bh[0] = m.get_param('Bh', (3 * D,), 0); o.H[0] = m.get_param('H', (B, D), 0)
o.Wy = m.get_param('Wy', (I, D)); o.By = m.get_param('By', (I,))
o.acc['Wy'] = m.get_param('acc_Wy', (I, D)); o.acc['By'] = m.get_param('acc_By', (I,))
o.global_step = N_STEP
t = N_STEP
cost, dbg = o.train_step(plan['in_idx'][t], plan['out_idx'][t], B, plan['reset'][t], return_debug=True)
m.train_steps(t, 1)
ld = int(m.get_debug('ldSc', (1,))[0])
ds = m.get_debug('scores', (B, ld))[:, :B + ns]
want = dbg['ds']
err = np.abs(ds - want)
print('ds max err', err.max(), 'max |ds|', np.abs(want).max(), 'cost', m.get_losses(t, 1)[0], cost)
idx = np.argsort(err.ravel())[::-1][:10]
cols = np.concatenate([plan['out_idx'][t], o.ST[t % o.generate_length]])
for e in idx:
    i, j = divmod(e, B + ns)
    print('row %d col %d item %d hip ds %.6e oracle ds %.6e | oracle s %.6e yhat %.6e | s_ii %.6e | row max yhat %.4e' % (
        i, j, cols[j], ds[i, j], want[i, j], dbg['s'][i, j], dbg['yhat'][i, j], dbg['s'][i, i], dbg['yhat'][i].max()))
