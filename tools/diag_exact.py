"""Diagnostic of the exact-replica mode: per-step costs of N virtual ranks against the oracle run as replicas (T steps)."""
import os, sys, copy
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from gru4rec_amd import _native
from test_gpu_parity import make_pair, random_plan
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
exact = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = dict(layers=(16,), loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, learning_rate=0.1, bpreg=1.0)
if len(sys.argv) > 4:
    kw['grad_cap'] = float(sys.argv[4])
I, B, ns = 40, 8, 16
pairs = [make_pair(I, B, ns, store_rows=T, seed=3, rank=r, nranks=N, sparse_exact=exact, **dict(kw)) for r in range(N)]
plans = [random_plan(I, B, T, seed=100 + r) for r in range(N)]
rng = np.random.RandomState(9)
for r, (o, m) in enumerate(pairs):
    o.ST = rng.randint(0, I, size=(T, ns)).astype(np.int64); o.generate_length = T
    m.set_sample_store(o.ST.astype(np.int32)); m.set_plan(plans[r])
oracles, models = [p[0] for p in pairs], [p[1] for p in pairs]
_native.virtual_train_steps(models, 0, T)
want = [[] for _ in range(N)]
for t in range(T):
    dense, sparse = [], []
    for r, o in enumerate(oracles):
        keep = copy.deepcopy({k: v for k, v in o.__dict__.items() if k not in ('dense_grad_hook', 'sparse_grad_hook')})
        cap = {}
        o.dense_grad_hook = lambda g, cap=cap: cap.setdefault('d', g)
        o.sparse_grad_hook = (lambda s, cap=cap: cap.setdefault('s', s)) if exact else None
        o.train_step(plans[r]['in_idx'][t], plans[r]['out_idx'][t], B, plans[r]['reset'][t], samples=o.ST[t])
        dense.append(cap['d']); sparse.append(cap.get('s'))
        o.__dict__.update(keep)
    avg = [(dense[0][j][0],) + tuple(None if dense[0][j][q] is None else sum(d[j][q] for d in dense) / N for q in range(1, 5)) for j in range(len(dense[0]))]
    if exact:
        names = [s[0] for s in sparse[0]]
        cat = [(nm, np.concatenate([sp[i][1] for sp in sparse]), np.concatenate([sp[i][2] for sp in sparse])) for i, nm in enumerate(names)]
    for r, o in enumerate(oracles):
        o.dense_grad_hook = lambda g, avg=avg: avg
        o.sparse_grad_hook = (lambda s, cat=cat: cat) if exact else None
        want[r].append(o.train_step(plans[r]['in_idx'][t], plans[r]['out_idx'][t], B, plans[r]['reset'][t], samples=o.ST[t]))
for r in range(N):
    got = models[r].get_losses(0, T)
    print('rank', r, 'costs got', got, 'want', np.array(want[r]))
    print('   dWx err %.3e  Wy err %.3e  acc_Wy err %.3e  By err %.3e' % (
        np.abs(models[r].get_param('Wx', (16, 48)) - oracles[r].Wx[0]).max(), np.abs(models[r].get_param('Wy', (I, 16)) - oracles[r].Wy).max(),
        np.abs(models[r].get_param('acc_Wy', (I, 16)) - oracles[r].acc['Wy']).max(), np.abs(models[r].get_param('By', (I,)) - oracles[r].By).max()))
if T == 1:
    r = 0
    dBy = np.abs(models[r].get_param('By', (I,)) - oracles[r].By)
    dWy = np.abs(models[r].get_param('Wy', (I, 16)) - oracles[r].Wy).max(axis=1)
    dA = np.abs(models[r].get_param('acc_Wy', (I, 16)) - oracles[r].acc['Wy']).max(axis=1)
    occ = []
    for q in range(N):
        occ.append(np.concatenate([plans[q]['in_idx'][0], plans[q]['out_idx'][0], oracles[q].ST[0]]))
    allocc = np.concatenate(occ)
    for it in range(I):
        pos = np.where(allocc == it)[0]
        flag = 'BAD' if (dBy[it] > 1e-5 or dWy[it] > 1e-5 or dA[it] > 1e-7) else 'ok '
        print(flag, 'item %2d  dBy %.2e dWy %.2e dAcc %.2e  occurrences (global K): %s' % (it, dBy[it], dWy[it], dA[it], list(pos)))
