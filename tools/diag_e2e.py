"""Diagnostic: HIP vs oracle in lock-step on the e2e workload; reports when / where they part."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gru4rec_amd import synth, _native
from gru4rec_amd.gru4rec import GRU4Rec
from oracle.model import OracleGRU4Rec
from oracle.scheduler import fit_schedule

data = synth.make_sessions(24000, n_items=2500, seed=17)
train, test = synth.train_test_split(data, test_frac=0.1)
P = dict(loss='bpr-max', final_act='elu-0.5', layers=[100], batch_size=128, n_sample=2048, constrained_embedding=True,
         learning_rate=0.1, bpreg=1.0, momentum=0.0, sample_alpha=0.75, n_epochs=1)
gru = GRU4Rec(**P)
gru.prepare(train.copy(), sample_store=2048 * 640)
m = gru._model
plan = gru._epoch_plan()
m.reset_hidden()
I = gru.n_items
o = OracleGRU4Rec(n_items=I, layers=(100,), batch_size=128, loss='bpr-max', final_act='elu-0.5', n_sample=2048,
                  constrained_embedding=True, learning_rate=0.1, bpreg=1.0, sample_alpha=0.75, dtype=np.float32, seed=gru.seed)
sup = np.bincount(train.assign(ItemIdx=gru.itemidmap[train.ItemId.values].values).ItemIdx.values, minlength=I)
o.set_popularity(sup)
o.make_sample_store(2048 * 640)
assert np.array_equal(m.get_sample_store(2048), o.ST)
assert np.array_equal(o.Wy, gru.Wy)
T = plan['T']
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 20
t = 0
while t < T:
    n = min(chunk, T - t)
    m.train_steps(t, n)
    got = m.get_losses(t, n)
    want = np.array([o.train_step(plan['in_idx'][k], plan['out_idx'][k], int(plan['M'][k]), plan['reset'][k]) for k in range(t, t + n)])
    rel = np.abs(got - want) / np.abs(want)
    wy = m.get_param('Wy', (I, 100)); acc = m.get_param('acc_Wy', (I, 100))
    dw = np.abs(wy - o.Wy); da = np.abs(acc - o.acc['Wy'])
    wh = m.get_param('Wh', (100, 100), 0)
    print('steps %4d-%4d  cost rel err max %.2e | Wy max abs err %.2e (row %d) rel %.2e | acc max err %.2e (row %d) | Wh err %.2e | M %d' % (
        t, t + n - 1, rel.max(), dw.max(), dw.max(axis=1).argmax(), (dw / (np.abs(o.Wy) + 1e-3)).max(), da.max(), da.max(axis=1).argmax(),
        np.abs(wh - o.Wh[0]).max(), plan['M'][t + n - 1]), flush=True)
    t += n
