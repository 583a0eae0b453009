"""Step-by-step diagnostic of the exact-replica mode: after EVERY step, per-item errors of rank 0's tables against the oracle replicas."""
import os, sys, copy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gru4rec_amd import _native
from test_gpu_parity import make_pair, random_plan
N, T = int(sys.argv[1]), int(sys.argv[2])
kw = dict(layers=(16,), loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, learning_rate=0.1, bpreg=1.0)
I, B, ns = 40, 8, 16
pairs = [make_pair(I, B, ns, store_rows=T, seed=3, rank=r, nranks=N, sparse_exact=1, **dict(kw)) for r in range(N)]
plans = [random_plan(I, B, T, seed=100 + r) for r in range(N)]
rng = np.random.RandomState(9)
for r, (o, m) in enumerate(pairs):
    o.ST = rng.randint(0, I, size=(T, ns)).astype(np.int64); o.generate_length = T
    m.set_sample_store(o.ST.astype(np.int32)); m.set_plan(plans[r])
oracles, models = [p[0] for p in pairs], [p[1] for p in pairs]
for t in range(T):
    _native.virtual_train_steps(models, t, 1)
    dense, sparse = [], []
    for r, o in enumerate(oracles):
        keep = copy.deepcopy({k: v for k, v in o.__dict__.items() if k not in ('dense_grad_hook', 'sparse_grad_hook')})
        cap = {}
        o.dense_grad_hook = lambda g, cap=cap: cap.setdefault('d', g)
        o.sparse_grad_hook = lambda s, cap=cap: cap.setdefault('s', s)
        o.train_step(plans[r]['in_idx'][t], plans[r]['out_idx'][t], B, plans[r]['reset'][t], samples=o.ST[t])
        dense.append(cap['d']); sparse.append(cap['s'])
        o.__dict__.update(keep)
    avg = [(dense[0][j][0],) + tuple(None if dense[0][j][q] is None else sum(d[j][q] for d in dense) / N for q in range(1, 5)) for j in range(len(dense[0]))]
    names = [s[0] for s in sparse[0]]
    cat = [(nm, np.concatenate([sp[i][1] for sp in sparse]), np.concatenate([sp[i][2] for sp in sparse])) for i, nm in enumerate(names)]
    want = []
    for r, o in enumerate(oracles):
        o.dense_grad_hook = lambda g, avg=avg: avg
        o.sparse_grad_hook = lambda s, cat=cat: cat
        want.append(o.train_step(plans[r]['in_idx'][t], plans[r]['out_idx'][t], B, plans[r]['reset'][t], samples=o.ST[t]))
    got = [float(models[r].get_losses(t, 1)[0]) for r in range(N)]
    print('step', t, 'costs got', got, 'want', [float(w) for w in want])
    r = 0
    dBy = np.abs(models[r].get_param('By', (I,)) - oracles[r].By)
    dWy = np.abs(models[r].get_param('Wy', (I, 16)) - oracles[r].Wy).max(axis=1)
    dA = np.abs(models[r].get_param('acc_Wy', (I, 16)) - oracles[r].acc['Wy']).max(axis=1)
    dAB = np.abs(models[r].get_param('acc_By', (I,)) - oracles[r].acc['By'])
    dH = np.abs(models[r].get_param('H', (B, 16)) - oracles[r].H[0]).max()
    dWx = np.abs(models[r].get_param('Wx', (16, 48)) - oracles[r].Wx[0]).max()
    print('   H err %.2e  Wx err %.2e' % (dH, dWx))
    allocc = np.concatenate([np.concatenate([plans[q]['in_idx'][t], plans[q]['out_idx'][t], oracles[q].ST[t]]) for q in range(N)])
    for it in range(I):
        if dBy[it] > 1e-5 or dWy[it] > 1e-5 or dA[it] > 1e-7 or dAB[it] > 1e-7:
            print('   BAD item %2d  dBy %.2e dWy %.2e dAcc %.2e dAccBy %.2e occurrences (global K): %s' % (it, dBy[it], dWy[it], dA[it], dAB[it], [int(x) for x in np.where(allocc == it)[0]]))
