"""Debug: one step of the exact-replica MEAN form (sparse_exact = 2) against the per-rank restatement, item by item."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from gru4rec_amd import _native
import test_gpu_exact_replicas as T

N, I, B, ns, steps = 3, 40, 8, 16, 1
kw, oracles, models, plans = T._replica_setup('bprmax_constrained', 2, N, I, B, ns, steps, shared_negatives=False)
_native.virtual_train_steps(models, 0, steps)
avg, sparse = T._capture(oracles, plans, 0, B)
o0 = oracles[0]
for i, (nm, _, _) in enumerate(sparse[0]):
    P0, A0 = getattr(o0, nm).copy(), o0.acc[nm].copy()
    dP = np.zeros_like(P0, dtype=np.float64); dA = np.zeros_like(A0, dtype=np.float64)
    touch = np.zeros(P0.shape[0], dtype=np.int64)
    for r, o in enumerate(oracles):
        setattr(o, nm, P0.copy()); o.acc[nm] = A0.copy()
        o._sparse_update(nm, sparse[r][i][1], sparse[r][i][2])
        dP += getattr(o, nm).astype(np.float64) - P0
        dA += o.acc[nm].astype(np.float64) - A0
        touch[np.unique(sparse[r][i][1])] += 1
    for o in oracles:
        setattr(o, nm, P0.copy()); o.acc[nm] = A0.copy()
    nq = np.maximum(touch, 1).reshape((-1,) + (1,) * (P0.ndim - 1))
    want_P, want_A = P0 + dP / nq, A0 + dA
    shape = P0.shape
    got_P = models[0].get_param(nm, shape).astype(np.float64)
    got_A = models[0].get_param('acc_' + nm, shape).astype(np.float64)
    print('==', nm)
    for it in range(I):
        occ = [(r, int((sparse[r][i][1] == it).sum())) for r in range(N)]
        eP = np.abs(got_P[it] - want_P[it]).max(); eA = np.abs(got_A[it] - want_A[it]).max()
        uP = np.abs(want_P[it] - P0[it]).max(); uA = np.abs(want_A[it] - A0[it]).max()
        # what the SUM form would give
        sP = np.abs(got_P[it] - (P0[it] + dP[it])).max()
        flag = 'BAD' if (eP > 1e-3 * uP + 1e-9 or eA > 1e-3 * uA + 1e-12) else 'ok '
        print('%s item %2d occ %s  touch %d  errP %.2e (upd %.2e; vs SUM-form %.2e)  errA %.2e (upd %.2e)' % (flag, it, occ, touch[it], eP, uP, sP, eA, uA))
