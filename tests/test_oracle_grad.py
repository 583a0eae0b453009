"""Oracle self-check: hand-derived backward + update rules vs torch.autograd (CPU, float64).

torch is a test-only dependency here (an independent differentiator of the forward formulas
gru4rec.py:193-248,471-496); it is not used by the product.
"""
import numpy as np
import pytest
import torch

from oracle.model import OracleGRU4Rec, EPS_LOSS


def torch_forward_cost(o, P, in_idx, Yp, M, masks):
    """Forward pass written directly from the reference formulas, differentiable w.r.t. the dict P."""
    B = o.batch_size
    if o.constrained_embedding:
        Sx = P['S'][:M]
        Sy = P['S'][M:]
    else:
        Sx = P['Sx']
        Sy = P['Sy']
    y = Sx if (masks['embed'] is None or o.onehot) else Sx * torch.tensor(masks['embed'])
    for i, D in enumerate(o.layers):
        H = torch.tensor(o.H[i][:M])
        # one-hot input: Sx are the gathered rows of Wx[0] (gru4rec.py:458-459)
        V = (Sx if (o.onehot and i == 0) else y @ P['Wx%d' % i]) + P['Bh%d' % i]
        rz = torch.sigmoid(V[:, D:] + H @ P['Wrz%d' % i])
        a = (H * rz[:, :D]) @ P['Wh%d' % i] + V[:, :D]
        kind, p0, p1 = o.hidden_act
        if kind == 'tanh':
            c = torch.tanh(a)
        elif kind == 'relu':
            c = torch.relu(a)
        elif kind == 'elu':
            c = torch.where(a >= 0, a, p0 * (torch.exp(a) - 1))
        else:
            c = a
        z = rz[:, D:]
        h = (1 - z) * H + z * c
        if masks['hidden'][i] is not None:
            h = h * torch.tensor(masks['hidden'][i])
        y = h
    s = y @ Sy.T + P['SBy'][None, :]
    if o.logq:
        lq = np.concatenate([o.lq_tgt[Yp[:M]], o.lq_smp[Yp[M:]]]).astype(np.float64)
        s = s - o.logq * torch.tensor(lq)[None, :]
    kind, p0, p1 = o.final_act
    if kind == 'softmax':
        e = torch.exp(s - s.max(dim=1, keepdim=True).values)
        yhat = e / e.sum(dim=1, keepdim=True)
    elif kind == 'softmax_logit':      # gru4rec.py:196-198
        x = s - s.max(dim=1, keepdim=True).values
        yhat = torch.log(torch.exp(x).sum(dim=1, keepdim=True)) - x
    elif kind == 'elu':
        yhat = torch.where(s >= 0, s, p0 * (torch.exp(s) - 1))
    elif kind == 'tanh':
        yhat = torch.tanh(s)
    elif kind == 'relu':
        yhat = torch.relu(s)
    else:
        yhat = s
    N = s.shape[1]
    diag = torch.diagonal(yhat)[:, None]
    n_out = M + o.n_sample
    sm = o.smoothing
    if o.loss == 'cross-entropy':      # gru4rec.py:225-230
        cost = (-torch.log(diag[:, 0] + EPS_LOSS)).sum() if not sm else \
            ((1.0 - (n_out / (n_out - 1)) * sm) * (-torch.log(diag[:, 0] + EPS_LOSS))
             + (sm / (n_out - 1)) * (-torch.log(yhat + EPS_LOSS)).sum(dim=1)).sum()
    elif o.loss == 'xe_logit':         # :231-236
        cost = diag[:, 0].sum() if not sm else \
            ((1.0 - (n_out / (n_out - 1)) * sm) * diag[:, 0] + (sm / (n_out - 1)) * yhat.sum(dim=1)).sum()
    elif o.loss == 'bpr':              # :237-238
        cost = (-torch.log(torch.sigmoid(diag - yhat))).sum()
    elif o.loss == 'top1':             # :242-244
        # exactly as written: a (M,) vector minus a (M, 1) column broadcasts to (M, M) before the sum
        cost = ((torch.sigmoid(-diag + yhat) + torch.sigmoid(yhat ** 2)).mean(dim=1)
                - torch.sigmoid(diag ** 2) / n_out).sum()
    else:
        hm = 1.0 - torch.eye(M, N, dtype=s.dtype)
        X = yhat * hm
        e = torch.exp(X - X.max(dim=1, keepdim=True).values) * hm
        p = e / e.sum(dim=1, keepdim=True)
        if o.loss == 'bpr-max':
            cost = (-torch.log((torch.sigmoid(diag - yhat) * p).sum(dim=1) + EPS_LOSS)
                    + o.bpreg * ((yhat ** 2) * p).sum(dim=1)).sum()
        else:
            cost = (p * (torch.sigmoid(-diag + yhat) + torch.sigmoid(yhat ** 2))).sum()
    return cost / B


CASES = [
    dict(loss='bpr-max', final_act='elu-0.5', constrained_embedding=True, layers=(12,), bpreg=0.7),
    dict(loss='bpr-max', final_act='linear', constrained_embedding=True, layers=(12,), dropout_p_hidden=0.3,
         dropout_p_embed=0.2),
    dict(loss='top1-max', final_act='tanh', constrained_embedding=True, layers=(8, 12)),
    dict(loss='cross-entropy', final_act='softmax', constrained_embedding=True, layers=(12,), logq=1.0),
    dict(loss='cross-entropy', final_act='softmax', constrained_embedding=False, embedding=10, layers=(12,)),
    dict(loss='bpr-max', final_act='relu', hidden_act='relu', constrained_embedding=False, embedding=6,
         layers=(8, 8)),
    dict(loss='bpr', final_act='linear', constrained_embedding=True, layers=(12,)),
    dict(loss='bpr-max', final_act='elu-0.5', layers=(12,)),                          # one-hot input (constructor default)
    dict(loss='cross-entropy', final_act='softmax', layers=(8, 12), dropout_p_hidden=0.2),   # one-hot, 2 layers
    dict(loss='top1', final_act='tanh', constrained_embedding=True, layers=(12,)),
    dict(loss='xe_logit', final_act='softmax_logit', constrained_embedding=True, layers=(12,), smoothing=0.1),
    dict(loss='xe_logit', final_act='softmax_logit', constrained_embedding=False, embedding=8, layers=(12,)),
    dict(loss='cross-entropy', final_act='softmax', constrained_embedding=True, layers=(12,), smoothing=0.2, logq=1.0),
]


@pytest.mark.parametrize('case', CASES)
@pytest.mark.parametrize('M', [8, 5])
def test_backward_matches_autograd(case, M):
    rng = np.random.RandomState(1)
    I, B, ns = 50, 8, 16
    o = OracleGRU4Rec(n_items=I, batch_size=B, n_sample=ns, dtype=np.float64, **case)
    o.set_popularity(rng.randint(1, 30, size=I))
    for i in range(len(o.layers)):
        o.H[i] = rng.randn(B, o.layers[i]) * 0.5
        o.Bh[i] = rng.randn(3 * o.layers[i]) * 0.1
    o.By = rng.randn(I) * 0.1
    in_idx = rng.randint(0, I, size=M)
    out_idx = rng.randint(0, I, size=M)
    samples = rng.randint(0, I, size=ns)
    reset = rng.rand(M) < 0.3
    Yp = np.concatenate([out_idx, samples])
    # masks fixed so that both sides see the same dropout
    from oracle import philox
    masks = {'embed': None, 'hidden': [None] * len(o.layers)}
    n_in = o.layers[-1] if o.constrained_embedding else (o.embedding or 3 * o.layers[0])
    if o.dropout_p_embed > 0:
        masks['embed'] = philox.dropout_mask(M, n_in, 1 - o.dropout_p_embed, 7, 0, 1).astype(np.float64)
    if o.dropout_p_hidden > 0:
        masks['hidden'] = [philox.dropout_mask(M, D, 1 - o.dropout_p_hidden, 7, 0, 2 + i).astype(np.float64)
                           for i, D in enumerate(o.layers)]
    P = {}
    if o.constrained_embedding:
        P['S'] = torch.tensor(o.Wy[np.concatenate([in_idx, Yp])], requires_grad=True)
    else:
        P['Sx'] = torch.tensor((o.Wx[0] if o.onehot else o.E)[in_idx], requires_grad=True)
        P['Sy'] = torch.tensor(o.Wy[Yp], requires_grad=True)
    P['SBy'] = torch.tensor(o.By[Yp], requires_grad=True)
    for i in range(len(o.layers)):
        for n in ('Wx', 'Wh', 'Wrz', 'Bh'):
            if n == 'Wx' and i == 0 and o.onehot:
                continue
            P['%s%d' % (n, i)] = torch.tensor(getattr(o, n)[i], requires_grad=True)
    cost_t = torch_forward_cost(o, P, in_idx, Yp, M, masks)
    cost_t.backward()
    cost, dbg = o.train_step(in_idx, out_idx, M, reset, samples=samples, masks=masks, return_debug=True)
    assert abs(cost - cost_t.item()) < 1e-12 * max(1, abs(cost))
    if o.constrained_embedding:
        g = P['S'].grad.numpy()
        np.testing.assert_allclose(dbg['dSx'], g[:M], rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(dbg['dSy'], g[M:], rtol=1e-9, atol=1e-13)
    else:
        np.testing.assert_allclose(dbg['dSx'], P['Sx'].grad.numpy(), rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(dbg['dSy'], P['Sy'].grad.numpy(), rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(dbg['dSBy'], P['SBy'].grad.numpy(), rtol=1e-9, atol=1e-13)
    for (i, dWx, dWh, dWrz, dBh) in dbg['dense_grads']:
        if dWx is not None:
            np.testing.assert_allclose(dWx, P['Wx%d' % i].grad.numpy(), rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(dWh, P['Wh%d' % i].grad.numpy(), rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(dWrz, P['Wrz%d' % i].grad.numpy(), rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(dBh, P['Bh%d' % i].grad.numpy(), rtol=1e-9, atol=1e-13)


def test_sparse_update_duplicate_semantics():
    """acc: last occurrence wins; param increments accumulate (gru4rec.py:336-338,431)."""
    o = OracleGRU4Rec(n_items=5, layers=(4,), batch_size=2, n_sample=0, constrained_embedding=True,
                      dtype=np.float64, learning_rate=0.5)
    o.Wy[:] = 1.0
    idx = np.array([3, 3, 1])
    g = np.array([[1.0] * 4, [2.0] * 4, [3.0] * 4])
    o._sparse_update('Wy', idx, g)
    assert np.allclose(o.acc['Wy'][3], 4.0)          # from the last occurrence (g=2)
    d1 = 0.5 * 1.0 / np.sqrt(1.0 + 1e-6)
    d2 = 0.5 * 2.0 / np.sqrt(4.0 + 1e-6)
    assert np.allclose(o.Wy[3], 1.0 - d1 - d2)
    assert np.allclose(o.Wy[1], 1.0 - 0.5 * 3.0 / np.sqrt(9.0 + 1e-6))
    assert np.allclose(o.Wy[0], 1.0)
