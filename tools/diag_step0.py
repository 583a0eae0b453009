import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gru4rec_amd import synth, _native
from gru4rec_amd.gru4rec import GRU4Rec
from oracle.model import OracleGRU4Rec
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
cost, dbg = o.train_step(plan['in_idx'][0], plan['out_idx'][0], B, plan['reset'][0], return_debug=True)
m.train_steps(0, 1)
ld = int(m.get_debug('ldSc', (1,))[0])
ds = m.get_debug('scores', (B, ld))[:, :B + ns]
want = dbg['ds']
err = np.abs(ds - want)
print('ds max err', err.max(), 'max |ds|', np.abs(want).max())
idx = np.argsort(err.ravel())[::-1][:12]
cols = np.concatenate([plan['out_idx'][0], o.ST[0]])
for e in idx:
    i, j = divmod(e, B + ns)
    print('row %d col %d item %d (target item of row %d) hip %.6e oracle %.6e | same-item cols: %s' % (i, j, cols[j], cols[i], ds[i, j], want[i, j], np.nonzero(cols == cols[j])[0][:8]))
hd = m.get_debug('hd0', (B, D)); print('hd err', np.abs(hd - dbg['caches'][0]['hd']).max())
ks = int(m.get_debug('ksplit', (1,))[0])
step = m.get_debug('dSBy', (ld,))[:B + ns]
g = np.asarray(dbg['dSBy'], dtype=np.float64); ws = 0.1 * g / np.sqrt(g * g + 1e-6)
e2 = np.abs(step - ws); print('dSBy step max err', e2.max(), 'at col', e2.argmax(), 'hip', step[e2.argmax()], 'want', ws[e2.argmax()], 'g', g[e2.argmax()])
by = m.get_param('By', (I,)); eb = np.abs(by - o.By); print('By err', eb.max(), 'item', eb.argmax(), 'hip', by[eb.argmax()], 'oracle', o.By[eb.argmax()], 'cols of that item', np.nonzero(cols == eb.argmax())[0])
