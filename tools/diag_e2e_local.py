"""Diagnostic: per-step (local) error of the HIP step against the oracle started from the SAME state."""
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
gru = GRU4Rec(**P)
gru.prepare(train.copy(), sample_store=2048 * 640)
m = gru._model
plan = gru._epoch_plan()
m.reset_hidden()
I, D, B = gru.n_items, 100, 128
o = OracleGRU4Rec(n_items=I, layers=(100,), batch_size=128, loss='bpr-max', final_act='elu-0.5', n_sample=2048,
                  constrained_embedding=True, learning_rate=0.1, bpreg=1.0, sample_alpha=0.75, dtype=np.float32, seed=gru.seed)
sup = np.bincount(train.assign(ItemIdx=gru.itemidmap[train.ItemId.values].values).ItemIdx.values, minlength=I)
o.set_popularity(sup)
o.make_sample_store(2048 * 640)

def sync():
    o.Wx[0] = m.get_param('Wx', (D, 3 * D), 0); o.Wh[0] = m.get_param('Wh', (D, D), 0); o.Wrz[0] = m.get_param('Wrz', (D, 2 * D), 0)
    o.Bh[0] = m.get_param('Bh', (3 * D,), 0); o.H[0] = m.get_param('H', (B, D), 0)
    o.Wy = m.get_param('Wy', (I, D)); o.By = m.get_param('By', (I,))
    o.acc['Wy'] = m.get_param('acc_Wy', (I, D)); o.acc['By'] = m.get_param('acc_By', (I,))
    o.acc['Wx'][0] = m.get_param('acc_Wx', (D, 3 * D), 0); o.acc['Wh'][0] = m.get_param('acc_Wh', (D, D), 0)
    o.acc['Wrz'][0] = m.get_param('acc_Wrz', (D, 2 * D), 0); o.acc['Bh'][0] = m.get_param('acc_Bh', (3 * D,), 0)

t = 0
import sys as _s
LIM = int(_s.argv[1]) if len(_s.argv) > 1 else 130
for probe in range(LIM):
    if probe > t:
        m.train_steps(t, probe - t)
        t = probe
    sync()
    o.global_step = t
    wy0, by0 = o.Wy.copy(), o.By.copy()
    wh0 = o.Wh[0].copy()
    want = o.train_step(plan['in_idx'][t], plan['out_idx'][t], int(plan['M'][t]), plan['reset'][t])
    m.train_steps(t, 1)
    got = m.get_losses(t, 1)[0]
    t += 1
    wy = m.get_param('Wy', (I, D)); by = m.get_param('By', (I,)); wh = m.get_param('Wh', (D, D), 0); acc = m.get_param('acc_Wy', (I, D))
    d_h, d_o = wy - wy0, o.Wy - wy0
    err = np.abs(d_h - d_o)
    row = err.max(axis=1).argmax()
    occ = np.concatenate([plan['in_idx'][t - 1], plan['out_idx'][t - 1], o.ST[(t - 1) % o.generate_length]])
    cnt = np.bincount(occ, minlength=I)
    scale = np.abs(d_o).max()
    if err.max() < 2e-6 and np.abs(by - o.By).max() < 5e-6: continue
    print('step %3d cost %.7f vs %.7f | dWy err max %.3e (row %d, count %d; |d| max %.3e) | acc err %.2e | By err %.2e | dWh err %.2e (|d| %.2e) | H err %.2e' % (
        t - 1, got, want, err.max(), row, cnt[row], scale, np.abs(acc - o.acc['Wy']).max(), np.abs(by - o.By).max(),
        np.abs((wh - wh0) - (o.Wh[0] - wh0)).max(), np.abs(o.Wh[0] - wh0).max(), np.abs(m.get_param('H', (B, D), 0) - o.H[0]).max()), flush=True)
    # per-count error profile
    rerr = err.max(axis=1)
    for lo, hi in ((1, 1), (2, 2), (3, 9), (10, 64), (65, 10000)):
        sel = (cnt >= lo) & (cnt <= hi)
        if sel.any():
            print('      count %d-%d: rows %d  max err %.3e  max |d| %.3e' % (lo, hi, sel.sum(), rerr[sel].max(), np.abs(d_o[sel]).max()))
    sel = cnt == 0
    print('      untouched rows: max |d hip| %.3e' % np.abs(d_h[sel]).max())
    bad = np.argsort(rerr)[::-1][:6]
    for r_ in bad:
        pos = np.nonzero(occ == r_)[0]
        print('      row %d err %.3e count %d positions %s  |d_o| %.3e |d_h| %.3e  acc err %.3e' % (r_, rerr[r_], cnt[r_], pos[:12], np.abs(d_o[r_]).max(), np.abs(d_h[r_]).max(), np.abs(acc[r_] - o.acc['Wy'][r_]).max()))
