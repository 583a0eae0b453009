"""CPU restatement of the GRU4Rec training / prediction step (TEST INFRASTRUCTURE).

This is the *oracle* for the HIP hot path: a NumPy restatement of what the reference's compiled
Theano function computes per mini-batch (forward `model()` gru4rec.py:433-506, losses :225-248,
backward = T.grad :383-384, Adagrad(+momentum) updates :330-340,382-432), of the weight
initialisation (:252-260,267-294), of the negative-sample store (:539-571 with the GPU
searchsorted semantics of custom_theano_ops.py:318-349) and of prediction/evaluation
(gru4rec.py:665-741, evaluation.py:47-147).

Pinning: the reference cannot be imported directly (Theano is not installable here), so the oracle
is pinned against golden vectors produced by executing the reference's own `gru4rec.py` on top of
a minimal Theano shim (`oracle/theano_shim`, `oracle/make_golden.py`, fixtures in `tests/golden`).

It must never be imported by the product path (`gru4rec_amd/`): only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg use it, as the checker.
"""
import numpy as np

from . import philox

EPS_LOSS = 1e-24      # gru4rec.py:230,241
EPS_ADAGRAD = 1e-6    # gru4rec.py:330


# ----------------------------------------------------------------------------- activations
def parse_act(name):
    """'elu-0.5' -> ('elu', 0.5, 0.0) ; mirrors gru4rec.py:144-161."""
    if name in ('linear', 'relu', 'tanh', 'softmax', 'softmax_logit'):
        return (name, 0.0, 0.0)
    if name.startswith('leaky-'):
        return ('leaky', float(name.split('-')[1]), 0.0)
    if name.startswith('elu-'):
        return ('elu', float(name.split('-')[1]), 0.0)
    if name.startswith('selu-'):
        p = [float(x) for x in name.split('-')[1:]]
        return ('selu', p[0], p[1])
    raise NotImplementedError(name)


def act_fwd(kind, p0, p1, x):
    """Element-wise activations gru4rec.py:189-223 (softmax handled separately)."""
    dt = x.dtype.type
    if kind == 'linear':
        return x
    if kind == 'relu':
        return np.maximum(x, dt(0))
    if kind == 'tanh':
        return np.tanh(x)
    if kind == 'leaky':
        return np.where(x >= 0, x, dt(p0) * x)
    if kind == 'elu':
        return np.where(x >= 0, x, dt(p0) * (np.exp(np.minimum(x, dt(0))) - dt(1)))
    if kind == 'selu':
        return dt(p0) * np.where(x >= 0, x, dt(p1) * (np.exp(np.minimum(x, dt(0))) - dt(1)))
    raise NotImplementedError(kind)


def act_bwd(kind, p0, p1, x, y):
    """d act / d x given input x and output y."""
    dt = x.dtype.type
    if kind == 'linear':
        return np.ones_like(x)
    if kind == 'relu':
        return (x > 0).astype(x.dtype)
    if kind == 'tanh':
        return dt(1) - y * y
    if kind == 'leaky':
        return np.where(x >= 0, dt(1), dt(p0))
    if kind == 'elu':
        return np.where(x >= 0, dt(1), y + dt(p0))
    if kind == 'selu':
        return np.where(x >= 0, dt(p0), y + dt(p0) * dt(p1))
    raise NotImplementedError(kind)


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x)) if x.dtype == np.float64 else \
        (np.float32(1.0) / (np.float32(1.0) + np.exp(-x))).astype(np.float32)


# ----------------------------------------------------------------------------- loss + final act
def softmax_rows(s, colmask):
    """softmax over the active columns (gru4rec.py:193-195)."""
    neg = np.where(colmask[None, :], s, -np.inf)
    m = neg.max(axis=1, keepdims=True)
    e = np.where(colmask[None, :], np.exp(s - m), 0).astype(s.dtype)
    return e / e.sum(axis=1, keepdims=True)


def final_act_fwd(kind, p0, p1, s, colmask):
    if kind == 'softmax':
        return softmax_rows(s, colmask)
    if kind == 'softmax_logit':      # gru4rec.py:196-198: log(sum exp(X - max)) - (X - max)
        neg = np.where(colmask[None, :], s, -np.inf)
        x = s - neg.max(axis=1, keepdims=True)
        e = np.where(colmask[None, :], np.exp(x), 0).astype(s.dtype)
        return (np.log(e.sum(axis=1, keepdims=True)) - x).astype(s.dtype)
    return act_fwd(kind, p0, p1, s)


def final_act_bwd(kind, p0, p1, s, yhat, dyhat, colmask):
    if kind == 'softmax':
        d = np.where(colmask[None, :], dyhat, 0).astype(s.dtype)
        inner = (d * yhat).sum(axis=1, keepdims=True)
        return yhat * (d - inner)
    if kind == 'softmax_logit':      # yhat_j = lse - x_j  =>  dx_k = softmax_k * sum_j d_j - d_k, softmax_k = exp(-yhat_k)
        d = np.where(colmask[None, :], dyhat, 0).astype(s.dtype)
        return (np.exp(-yhat) * d.sum(axis=1, keepdims=True) - d).astype(s.dtype)
    return dyhat * act_bwd(kind, p0, p1, s, yhat)


def softplus(x):
    return np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))


def loss_fwd_bwd(loss, yhat, M, diag_cols, colmask, bpreg, smoothing=0.0):
    """Return (sum of per-row losses, d/d yhat).  Rows 0..M-1; row i's positive is column diag_cols[i].

    softmax_neg restates gru4rec.py:199-203: the positive is masked out *after* being zeroed, the row
    max therefore includes the zeroed entry.  bpr_max :239-241, top1_max :245-248, cross_entropy :225-230.
    The derivative goes through the softmax weights as T.grad does.
    """
    dt = yhat.dtype.type
    rows = np.arange(M)
    hm = np.ones_like(yhat)
    hm[rows, diag_cols] = 0
    hm = hm * colmask[None, :].astype(yhat.dtype)
    ydiag = yhat[rows, diag_cols][:, None]
    cm = colmask[None, :].astype(yhat.dtype)
    n_out = dt(colmask.sum())
    if loss in ('cross-entropy', 'xe_logit'):
        # cross_entropy :225-230 / cross_entropy_logits :231-236, label smoothing included
        lg = loss == 'xe_logit'
        wd = dt(1) - n_out / (n_out - dt(1)) * dt(smoothing) if smoothing else dt(1)
        wa = dt(smoothing) / (n_out - dt(1)) if smoothing else dt(0)
        l_all = yhat if lg else -np.log(yhat + dt(EPS_LOSS))
        dl_all = np.ones_like(yhat) if lg else -dt(1) / (yhat + dt(EPS_LOSS))
        L = wd * l_all[rows, diag_cols] + wa * (l_all * cm).sum(axis=1)
        d = wa * dl_all * cm
        d[rows, diag_cols] += wd * dl_all[rows, diag_cols]
        return L.sum(dtype=yhat.dtype), d.astype(yhat.dtype)
    if loss == 'bpr':
        # bpr :237-238: -log sigmoid(yd - y_j) summed over ALL columns (the diagonal contributes log 2, no gradient)
        L = (softplus(yhat - ydiag) * cm).sum(axis=1)
        d = sigmoid(yhat - ydiag) * hm
        d[rows, diag_cols] = -d.sum(axis=1)
        return L.sum(dtype=yhat.dtype), d.astype(yhat.dtype)
    if loss == 'top1':
        # top1 :242-244: mean_j(sigmoid(y_j - yd) + sigmoid(y_j^2)) - sigmoid(yd^2) / n.  As written in the reference the
        # mean is a vector (M,) and the subtracted term a column (M, 1) (gpu_diag(..., keepdims=True)), so the
        # difference broadcasts to (M, M) before T.sum: the reference's cost is M times the per-row formula.  Kept.
        u = sigmoid(yhat - ydiag)
        q = sigmoid(yhat * yhat)
        L = (((u + q) * hm).sum(axis=1) + dt(0.5)) / n_out
        d = (u * (dt(1) - u) + dt(2) * yhat * q * (dt(1) - q)) * hm / n_out
        d[rows, diag_cols] = -((u * (dt(1) - u)) * hm).sum(axis=1) / n_out
        return dt(M) * L.sum(dtype=yhat.dtype), (dt(M) * d).astype(yhat.dtype)
    if loss in ('bpr-max', 'top1-max'):
        X = yhat * hm
        Xm = np.where(colmask[None, :], X, -np.inf)
        mx = Xm.max(axis=1, keepdims=True)
        e = (np.exp(X - mx) * hm).astype(yhat.dtype)
        p = e / e.sum(axis=1, keepdims=True)
        if loss == 'bpr-max':
            sg = sigmoid(ydiag - yhat)
            A = (sg * p).sum(axis=1, keepdims=True)
            Q = (yhat * yhat * p).sum(axis=1, keepdims=True)
            L = -np.log(A[:, 0] + dt(EPS_LOSS)) + dt(bpreg) * Q[:, 0]
            sgp = sg * (dt(1) - sg)
            d = -p * (sg - sgp - A) / (A + dt(EPS_LOSS)) + dt(bpreg) * p * (dt(2) * yhat + yhat * yhat - Q)
            d = d * hm
            d[rows, diag_cols] = -((sgp * p).sum(axis=1)) / (A[:, 0] + dt(EPS_LOSS))
            return L.sum(dtype=yhat.dtype), d.astype(yhat.dtype)
        else:
            u = sigmoid(yhat - ydiag)
            q = sigmoid(yhat * yhat)
            t = u + q
            T = (p * t).sum(axis=1, keepdims=True)
            L = T[:, 0]
            d = p * (t - T) + p * (u * (dt(1) - u) + dt(2) * yhat * q * (dt(1) - q))
            d = d * hm
            d[rows, diag_cols] = -((p * u * (dt(1) - u)).sum(axis=1))
            return L.sum(dtype=yhat.dtype), d.astype(yhat.dtype)
    raise NotImplementedError(loss)


# ----------------------------------------------------------------------------- the model
class OracleGRU4Rec:
    """NumPy GRU4Rec with the reference's semantics.  `dtype` float32 mirrors floatX; float64 = truth."""

    def __init__(self, n_items, layers=(100,), batch_size=32, loss='bpr-max', final_act='linear',
                 hidden_act='tanh', n_sample=2048, sample_alpha=0.75, learning_rate=0.1, momentum=0.0,
                 lmbd=0.0, bpreg=1.0, logq=0.0, dropout_p_hidden=0.0, dropout_p_embed=0.0,
                 constrained_embedding=False, embedding=0, sigma=0.0, init_as_normal=False,
                 dtype=np.float32, seed=12345, smoothing=0.0, adapt='adagrad', adapt_params=(), grad_cap=0.0):
        self.n_items = n_items
        self.layers = list(layers)
        self.batch_size = batch_size
        self.loss = loss
        self.final_act = parse_act(final_act)
        self.hidden_act = parse_act(hidden_act)
        self.n_sample = n_sample
        self.sample_alpha = sample_alpha
        self.learning_rate = learning_rate
        self.momentum = momentum
        self.lmbd = lmbd
        self.bpreg = bpreg
        self.smoothing = smoothing
        self.adapt = adapt                    # None | 'adagrad' | 'rmsprop' | 'adadelta' | 'adam' (gru4rec.py:392-399)
        self.adapt_params = list(adapt_params)
        self.grad_cap = grad_cap
        if adapt == 'adadelta':
            self.learning_rate = learning_rate = 1.0      # gru4rec.py:362-364
        self.logq = logq
        self.dropout_p_hidden = dropout_p_hidden
        self.dropout_p_embed = dropout_p_embed
        self.constrained_embedding = constrained_embedding
        self.embedding = embedding
        self.sigma = sigma
        self.init_as_normal = init_as_normal
        self.dtype = np.dtype(dtype)
        self.seed = seed
        self.global_step = 0
        self.ST = None
        # no embedding at all (the constructor default): layer 0 reads rows of Wx[0] (I x 3D), gru4rec.py:457-470
        self.onehot = (not constrained_embedding) and (not embedding)
        self._init_weights()

    # -- gru4rec.py:252-260
    def _init_matrix(self, shape):
        sigma = self.sigma if self.sigma != 0 else np.sqrt(6.0 / (shape[0] + shape[1]))
        if self.init_as_normal:
            return (np.random.randn(*shape) * sigma).astype(np.float32)
        return (np.random.rand(*shape) * sigma * 2 - sigma).astype(np.float32)

    # -- gru4rec.py:267-294 (same draw order after np.random.seed(42))
    def _init_weights(self):
        np.random.seed(42)
        L = self.layers
        self.E = None
        if self.constrained_embedding:
            n_features = L[-1]
        elif self.embedding:
            self.E = self._init_matrix((self.n_items, self.embedding))
            n_features = self.embedding
        else:
            n_features = self.n_items        # gru4rec.py:278
        self.Wx, self.Wh, self.Wrz, self.Bh, self.H = [], [], [], [], []
        for i in range(len(L)):
            n_in = L[i - 1] if i > 0 else n_features
            self.Wx.append(np.hstack([self._init_matrix((n_in, L[i])) for _ in range(3)]))
            self.Wh.append(self._init_matrix((L[i], L[i])))
            self.Wrz.append(np.hstack([self._init_matrix((L[i], L[i])) for _ in range(2)]))
            self.Bh.append(np.zeros(3 * L[i], dtype=np.float32))
            self.H.append(np.zeros((self.batch_size, L[i]), dtype=np.float32))
        self.Wy = self._init_matrix((self.n_items, L[-1]))
        self.By = np.zeros(self.n_items, dtype=np.float32)
        self._cast()
        self._init_opt_state()

    def _cast(self):
        dt = self.dtype
        if self.E is not None:
            self.E = self.E.astype(dt)
        for n in ('Wx', 'Wh', 'Wrz', 'Bh', 'H'):
            setattr(self, n, [a.astype(dt) for a in getattr(self, n)])
        self.Wy = self.Wy.astype(dt)
        self.By = self.By.astype(dt)

    def _init_opt_state(self):
        z = np.zeros_like
        self.acc = {'Wy': z(self.Wy), 'By': z(self.By)}
        self.vel = {'Wy': z(self.Wy), 'By': z(self.By)}
        if self.E is not None:
            self.acc['E'] = z(self.E)
            self.vel['E'] = z(self.E)
        for n in ('Wx', 'Wh', 'Wrz', 'Bh'):
            self.acc[n] = [z(a) for a in getattr(self, n)]
            self.vel[n] = [z(a) for a in getattr(self, n)]
        # second statistic (adadelta: upd, adam: meang) and adam's step counter, same shapes
        self.acc2 = {k: ([z(a) for a in v] if isinstance(v, list) else z(v)) for k, v in self.acc.items()}
        self.cnt = {k: ([z(a) for a in v] if isinstance(v, list) else z(v)) for k, v in self.acc.items()}

    # -- gru4rec.py:539-545 ; P is float32 on the device (:556)
    def set_popularity(self, support):
        support = np.asarray(support, dtype=np.float64)
        self.P0 = support.astype(np.float32)
        pop = support ** self.sample_alpha
        pop = pop.cumsum() / pop.sum()
        pop[-1] = 1
        self.pop64 = pop                      # the CPU store samples from the float64 table (gru4rec.py:507-514)
        self.P = pop.astype(np.float32)
        # logQ tables (gru4rec.py:495): log(P0) for in-batch targets, log(P0**alpha) for sampled negatives
        self.lq_tgt = np.log(self.P0).astype(np.float32)
        self.lq_smp = np.log(self.P0 ** np.float32(self.sample_alpha)).astype(np.float32)

    # -- custom_theano_ops.py:318-349 (upper bound with end clamps)
    @staticmethod
    def searchsorted_gpu(P, x):
        out = np.searchsorted(P, x, side='right').astype(np.int64)
        out[x > P[-1]] = len(P)
        out[x <= P[0]] = 0
        return out

    def make_sample_store(self, sample_store, store_type='gpu'):
        """gru4rec.py:546-566: ST[generate_length, n_sample]; uniforms from Philox instead of MRG (store_type 'gpu'), or the
        reference's own host sampler on NumPy's global stream (store_type 'cpu', gru4rec.py:507-514,551-554).  A store that cannot hold
        two rows means "no store": a fresh row of negatives is drawn for every step (gru4rec.py:548-550,614-615), which is a
        one-row store refilled before every step."""
        self.generate_length = sample_store // self.n_sample if self.n_sample else 0
        self.n_refills = 0
        self.store_type = store_type
        if self.n_sample:
            self.generate_length = max(self.generate_length, 1)
            self._refill()
        else:
            self.ST = None

    def _refill(self):
        n = self.generate_length * self.n_sample
        if getattr(self, 'store_type', 'gpu') == 'cpu':      # generate_neg_samples, gru4rec.py:507-514 (side='left', float64 table)
            if self.sample_alpha:
                smp = np.searchsorted(self.pop64, np.random.rand(n))
            else:
                smp = np.random.choice(self.n_items, size=n)
            self.ST = smp.reshape(self.generate_length, self.n_sample).astype(np.int32)
            self.n_refills += 1
            return
        u = philox.uniform_block(n, self.seed, self.n_refills, 0, philox.STREAM_SAMPLE)
        self.ST = self.searchsorted_gpu(self.P, u).reshape(self.generate_length, self.n_sample).astype(np.int32)
        self.n_refills += 1

    def next_samples(self):
        """Row of negatives for this step: pointer semantics of gru4rec.py:617-621 / STI :583."""
        if self.ST is None:
            return np.zeros(0, dtype=np.int64)
        k = self.global_step
        if k > 0 and k % self.generate_length == 0:
            self._refill()
        return self.ST[k % self.generate_length].astype(np.int64)

    # ------------------------------------------------------------------ forward pieces
    def _gru_fwd(self, i, y, H, rows=None):
        D = self.layers[i]
        # rows: one-hot input, the "product" with Wx[0] is a row gather (gru4rec.py:458-459)
        V = (self.Wx[i][rows] if rows is not None else y @ self.Wx[i]) + self.Bh[i]
        G = H @ self.Wrz[i]
        rz = sigmoid(V[:, D:] + G)
        r, z = rz[:, :D], rz[:, D:]
        a = (H * r) @ self.Wh[i] + V[:, :D]
        c = act_fwd(*self.hidden_act, a)
        h = (1 - z) * H + z * c
        return dict(y=y, H=H, r=r, z=z, a=a, c=c, h=h.astype(self.dtype))

    def train_step(self, in_idx, out_idx, M, reset, samples=None, masks=None, return_debug=False):
        """One call of the reference's `train_function(X, Y, M, R)` (gru4rec.py:576-584,623).

        `samples` overrides the sample store row; `masks` = dict(embed=.., hidden=[..]) overrides dropout.
        Returns cost (= sum of row losses / batch_size, gru4rec.py:577).
        """
        dt = self.dtype.type
        B = self.batch_size
        L = self.layers
        if M == 0:
            # padding step of a data-parallel plan (NOT in the reference: DESIGN.md section 7): nothing is active on this rank, it
            # contributes zero dense gradients to the all-reduce, applies the reduced ones and consumes its row of negatives
            self.next_samples()
            zeros = [(i, None if (self.onehot and i == 0) else np.zeros_like(self.Wx[i]), np.zeros_like(self.Wh[i]),
                      np.zeros_like(self.Wrz[i]), np.zeros_like(self.Bh[i])) for i in reversed(range(len(L)))]
            if getattr(self, 'dense_grad_hook', None) is not None:
                zeros = self.dense_grad_hook(zeros)
            for (i, dWx, dWh, dWrz, dBh) in zeros:
                if dWx is not None:
                    self._dense_update('Wx', i, dWx)
                self._dense_update('Wh', i, dWh)
                self._dense_update('Wrz', i, dWrz)
                self._dense_update('Bh', i, dBh)
            self.global_step += 1
            return dt(0)
        in_idx = np.asarray(in_idx, dtype=np.int64)[:M]
        out_idx = np.asarray(out_idx, dtype=np.int64)[:M]
        reset = np.asarray(reset).astype(bool).reshape(-1)[:M]
        if samples is None:
            samples = self.next_samples()
        samples = np.asarray(samples, dtype=np.int64)
        ns = len(samples)
        Yp = np.concatenate([out_idx, samples])          # gru4rec.py:436-437
        N = M + ns
        step = self.global_step
        # ---- gather (gru4rec.py:438-456,480-489)
        if self.constrained_embedding:
            Xc = np.concatenate([in_idx, Yp])
            S = self.Wy[Xc]
            Sx, Sy = S[:M], S[M:]
        elif self.onehot:
            Sx = self.Wx[0][in_idx]          # gru4rec.py:458 (the rows already are x * Wx[0]; no embedding dropout)
            Sy = self.Wy[Yp]
        else:
            Sx = self.E[in_idx]
            Sy = self.Wy[Yp]
        SBy = self.By[Yp]
        # ---- dropout masks
        if masks is None:
            masks = {'embed': None, 'hidden': [None] * len(L)}
            if self.dropout_p_embed > 0 and not self.onehot:
                masks['embed'] = philox.dropout_mask(M, Sx.shape[1], 1 - self.dropout_p_embed, self.seed, step,
                                                     philox.STREAM_DROP_EMBED).astype(self.dtype)
            if self.dropout_p_hidden > 0:
                for i in range(len(L)):
                    masks['hidden'][i] = philox.dropout_mask(M, L[i], 1 - self.dropout_p_hidden, self.seed, step,
                                                             philox.STREAM_DROP_HIDDEN + i).astype(self.dtype)
        y = Sx if (masks['embed'] is None or self.onehot) else Sx * masks['embed']
        # ---- GRU layers (gru4rec.py:471-479)
        caches = []
        for i in range(len(L)):
            cch = self._gru_fwd(i, y, self.H[i][:M], rows=in_idx if (self.onehot and i == 0) else None)
            hd = cch['h'] if masks['hidden'][i] is None else cch['h'] * masks['hidden'][i]
            cch['hd'] = hd
            caches.append(cch)
            y = hd
        # ---- scoring (gru4rec.py:493-496)
        s = y @ Sy.T + SBy[None, :]
        if self.logq:
            lq = np.concatenate([self.lq_tgt[Yp[:M]], self.lq_smp[Yp[M:]]]).astype(self.dtype)
            s = s - dt(self.logq) * lq[None, :]
        s = s.astype(self.dtype)
        # Scores on the KINK of a piecewise final activation (relu / leaky / elu / selu: T.switch(X >= 0), gru4rec.py:189-223): a
        # score whose magnitude is within the fp32 rounding of its own terms (|s| <= kink_ulps * eps32 * (sum_k |h_k w_k| + |b|)) is
        # positive in one summation order and negative in another, and its derivative takes either slope.  Test infrastructure:
        # `kink_flip` makes this oracle take the OTHER slope on exactly those elements, so that a test can bound the rows they touch
        # by the two runs instead of leaving them out (tests/test_gpu_parity.py: compare_params).
        kmask = None
        if str(self.final_act[0]) in ('relu', 'leaky', 'elu', 'selu') and (return_debug or getattr(self, 'kink_flip', False)):
            terms = np.abs(y).astype(np.float64) @ np.abs(Sy).astype(np.float64).T + np.abs(SBy).astype(np.float64)[None, :]
            if self.logq:
                terms = terms + abs(float(self.logq)) * np.abs(lq).astype(np.float64)[None, :]
            kmask = np.abs(s).astype(np.float64) <= float(getattr(self, 'kink_ulps', 4.0)) * float(np.finfo(np.float32).eps) * terms
        colmask = np.ones(N, dtype=bool)
        yhat = final_act_fwd(*self.final_act, s, colmask).astype(self.dtype)
        diag = np.arange(M)
        Lsum, dyhat = loss_fwd_bwd(self.loss, yhat, M, diag, colmask, self.bpreg, self.smoothing)
        cost = dt(Lsum) / dt(B)
        # ---- backward (T.grad, gru4rec.py:383-384)
        s_bwd = s
        if kmask is not None and getattr(self, 'kink_flip', False) and kmask.any():
            tiny = dt(np.finfo(np.float32).tiny)
            s_bwd = np.where(kmask, np.where(s >= 0, -tiny, tiny), s).astype(self.dtype)      # the other branch of the switch
        ds = (final_act_bwd(*self.final_act, s_bwd, yhat, dyhat, colmask) / dt(B)).astype(self.dtype)
        dSy = ds.T @ y
        dSBy = ds.sum(axis=0)
        dtop = ds @ Sy
        dense_grads = []
        dy = dtop
        for i in reversed(range(len(L))):
            cch = caches[i]
            D = L[i]
            dh = dy if masks['hidden'][i] is None else dy * masks['hidden'][i]
            H, r, z, c = cch['H'], cch['r'], cch['z'], cch['c']
            dz = dh * (c - H)
            dc = dh * z
            da = dc * act_bwd(*self.hidden_act, cch['a'], c)
            Hr = H * r
            dWh = Hr.T @ da
            dr = (da @ self.Wh[i].T) * H
            drp = dr * r * (1 - r)
            dzp = dz * z * (1 - z)
            dWrz = H.T @ np.hstack([drp, dzp])
            dV = np.hstack([da, drp, dzp]).astype(self.dtype)
            dBh = dV.sum(axis=0)
            if self.onehot and i == 0:
                dWx, dy = None, dV            # the gathered rows of Wx[0] receive dV itself (sparse update below)
            else:
                dWx = (cch['y'].T @ dV).astype(self.dtype)
                dy = dV @ self.Wx[i].T
            dense_grads.append((i, dWx, dWh.astype(self.dtype), dWrz.astype(self.dtype), dBh.astype(self.dtype)))
        dSx = dy if (masks['embed'] is None or self.onehot) else dy * masks['embed']
        dbg = None
        if return_debug:
            dbg = dict(Sx=Sx, Sy=Sy, s=s, yhat=yhat, ds=ds, dSy=dSy, dSBy=dSBy, dtop=dtop, dSx=dSx,
                       caches=caches, dense_grads=dense_grads, Yp=Yp, cost=cost,
                       kink_cols=(np.where(kmask.any(axis=0))[0] if kmask is not None else np.zeros(0, dtype=np.int64)))
        # data-parallel hook (not in the reference): all-reduce of the dense GRU gradients across ranks
        if getattr(self, 'dense_grad_hook', None) is not None:
            dense_grads = self.dense_grad_hook(dense_grads)
        # ---- global gradient-norm clipping over the dense gradients and the per-occurrence sparse gradients (gru4rec.py:386-389)
        if self.grad_cap > 0:
            sq = dt(0)
            for (i, dWx, dWh, dWrz, dBh) in dense_grads:
                for g in (dWx, dWh, dWrz, dBh):
                    if g is not None:
                        sq = sq + (g.astype(self.dtype) ** 2).sum(dtype=self.dtype)
            for g in (dSx, dSy, dSBy):
                sq = sq + (g.astype(self.dtype) ** 2).sum(dtype=self.dtype)
            norm = np.sqrt(sq)
            if norm >= self.grad_cap:
                c = dt(self.grad_cap) / norm
                dense_grads = [(i,) + tuple(None if g is None else (g * c).astype(self.dtype) for g in gs) for (i, *gs) in dense_grads]
                dSx, dSy, dSBy = (dSx * c).astype(self.dtype), (dSy * c).astype(self.dtype), (dSBy * c).astype(self.dtype)
        # ---- updates: everything reads pre-step values (Theano updates are simultaneous)
        newH = []
        for i in range(len(L)):
            hn = np.where(reset[:, None], dt(0), caches[i]['hd']).astype(self.dtype)
            Hfull = self.H[i].copy()
            Hfull[:M] = hn
            newH.append(Hfull)
        for (i, dWx, dWh, dWrz, dBh) in dense_grads:
            if dWx is not None:
                self._dense_update('Wx', i, dWx)
            self._dense_update('Wh', i, dWh)
            self._dense_update('Wrz', i, dWrz)
            self._dense_update('Bh', i, dBh)
        if self.constrained_embedding:
            sparse = [('Wy', Xc, np.vstack([dSx, dSy]).astype(self.dtype))]
        else:
            sparse = [('Wx0' if self.onehot else 'E', in_idx, dSx.astype(self.dtype)), ('Wy', Yp, dSy.astype(self.dtype))]
        sparse.append(('By', Yp, dSBy.astype(self.dtype)))
        # data-parallel hook (not in the reference; the product's exact-replica mode, g4r_config::sparse_exact): the per-occurrence
        # (index, gradient row) lists of ALL ranks, concatenated in rank order, are applied on every replica with the reference's
        # duplicate semantics -- the hook receives this rank's lists and returns the lists to apply
        if getattr(self, 'sparse_grad_hook', None) is not None:
            sparse = self.sparse_grad_hook(sparse)
        for (name, idx, g) in sparse:
            self._sparse_update(name, idx, g)
        self.H = newH
        self.global_step += 1
        if return_debug:
            return cost, dbg
        return cost

    # -- learning-rate adaptation of a dense gradient: adagrad :330-334, rmsprop :366-381, adadelta :341-365, adam :300-329
    def _adapt_dense(self, g, acc, acc2, cnt):
        """Returns (scaled gradient, new acc, new acc2, new cnt)."""
        dt = self.dtype.type
        eps = dt(EPS_ADAGRAD)
        if self.adapt == 'adagrad':
            an = acc + g * g
            return g / np.sqrt(an + eps), an, acc2, cnt
        if self.adapt == 'rmsprop':
            v1 = dt(self.adapt_params[0])
            an = v1 * acc + (dt(1) - v1) * g * g
            return g / np.sqrt(an + eps), an, acc2, cnt
        if self.adapt == 'adadelta':
            v1 = dt(self.adapt_params[0])
            an = v1 * acc + (dt(1) - v1) * g * g
            gs = (acc2 + eps) / (an + eps)
            un = v1 * acc2 + (dt(1) - v1) * gs * g * g
            return g * np.sqrt(gs), an, un, cnt
        if self.adapt == 'adam':
            v1, v3 = dt(self.adapt_params[0]), dt(self.adapt_params[1])
            an = v3 * acc + (dt(1) - v3) * g * g
            mn = v1 * acc2 + (dt(1) - v1) * g
            cn = cnt + dt(1)
            corr = dt(1) - v1 ** cn                       # both moments are corrected with v1 (:329)
            return (mn / corr) / (np.sqrt(an / corr) + eps), an, mn, cn
        return g, acc, acc2, cnt                        # adapt=None: plain SGD

    # -- gru4rec.py:390-406
    def _dense_update(self, name, i, g):
        dt = self.dtype.type
        p = getattr(self, name)[i]
        gs, an, a2, cn = self._adapt_dense(g, self.acc[name][i], self.acc2[name][i], self.cnt[name][i])
        self.acc[name][i] = an.astype(self.dtype)
        self.acc2[name][i] = a2.astype(self.dtype)
        self.cnt[name][i] = cn.astype(self.dtype)
        lr = dt(self.learning_rate)
        if self.momentum > 0:
            v = self.vel[name][i]
            v2 = dt(self.momentum) * v - lr * (gs + dt(self.lmbd) * p)
            self.vel[name][i] = v2.astype(self.dtype)
            getattr(self, name)[i] = (p + v2).astype(self.dtype)
        else:
            getattr(self, name)[i] = (p * (dt(1) - lr * dt(self.lmbd)) - lr * gs).astype(self.dtype)

    # -- gru4rec.py:407-431 with adagrad(sample_idx) :335-340
    def _sparse_update(self, name, idx, g):
        """Per-occurrence scaling from the pre-step accumulator; accumulator/velocity: the LAST occurrence
        of a duplicated index wins (NumPy/CPU `set_subtensor` semantics; on the GPU the reference's winner is
        unspecified); parameter increments of duplicates accumulate in occurrence order (`inc_subtensor`)."""
        dt = self.dtype.type
        if name == 'Wx0':       # one-hot input: Wx[0] is the sparse-updated (I, 3D) table (gru4rec.py:467-469,578)
            P, acc, vel = self.Wx[0], self.acc['Wx'][0], self.vel['Wx'][0]
        else:
            P, acc, vel = getattr(self, name), self.acc[name], self.vel[name]
        if name == 'Wx0':
            acc2, cnt = self.acc2['Wx'][0], self.cnt['Wx'][0]
        else:
            acc2, cnt = self.acc2[name], self.cnt[name]
        eps = dt(EPS_ADAGRAD)
        if self.adapt == 'adagrad':
            acc_new = acc[idx] + g * g
            gs = g / np.sqrt(acc_new + eps)
            acc[idx] = acc_new                                  # set_subtensor: last write wins (:336-338)
        elif self.adapt in ('rmsprop', 'adadelta', 'adam'):
            # the "accurate" duplicate handling of the reference (:321-326,349-358,373-378): the statistic of an index is
            # decayed once and receives the contribution of every occurrence; all occurrences see the final value
            v1 = dt(self.adapt_params[0])
            va = dt(self.adapt_params[1]) if self.adapt == 'adam' else v1
            uniq = np.unique(idx)
            acc_pre2 = acc2[idx].copy()
            acc[uniq] = acc[uniq] * va
            np.add.at(acc, idx, ((dt(1) - va) * g * g).astype(self.dtype))
            acc_new = acc[idx]
            if self.adapt == 'rmsprop':
                gs = g / np.sqrt(acc_new + eps)
            elif self.adapt == 'adadelta':
                sc = (acc_pre2 + eps) / (acc_new + eps)
                acc2[uniq] = acc2[uniq] * v1
                np.add.at(acc2, idx, ((dt(1) - v1) * sc * g * g).astype(self.dtype))
                gs = g * np.sqrt(sc)
            else:
                # adam, sparse branch as written: the mean accumulator also receives grad**2 (:325) and both moments are
                # bias-corrected with v1 (:329); the result does not depend on the occurrence's own gradient
                acc2[uniq] = acc2[uniq] * v1
                np.add.at(acc2, idx, ((dt(1) - v1) * g * g).astype(self.dtype))
                cn = cnt[idx] + dt(1)
                cnt[idx] = cn
                corr = dt(1) - v1 ** cn
                gs = (acc2[idx] / corr) / (np.sqrt(acc_new / corr) + eps)
        else:
            gs = g
        gs = gs.astype(self.dtype)
        lr = dt(self.learning_rate)
        if self.lmbd > 0:
            delta = lr * (gs + dt(self.lmbd) * P[idx])
        else:
            delta = lr * gs
        if self.momentum > 0:
            v2 = dt(self.momentum) * vel[idx] - delta
            vel[idx] = v2                                   # last write wins
            np.add.at(P, idx, v2.astype(self.dtype))        # accumulates, occurrence order
        else:
            np.add.at(P, idx, (-delta).astype(self.dtype))

    # ------------------------------------------------------------------ prediction (gru4rec.py:665-741)
    def predict_step(self, H_list, in_idx, item_idx=None):
        """Forward only (`predict=True`): no dropout, no logQ, no reset switch.  Returns (scores[M, n], new H)."""
        in_idx = np.asarray(in_idx, dtype=np.int64)
        y = self.Wy[in_idx] if self.constrained_embedding else (None if self.onehot else self.E[in_idx])
        newH = []
        for i in range(len(self.layers)):
            cch = self._gru_fwd(i, y, H_list[i], rows=in_idx if (self.onehot and i == 0) else None)
            y = cch['h']
            newH.append(y)
        if item_idx is None:
            Sy, SBy = self.Wy, self.By
        else:
            Sy, SBy = self.Wy[item_idx], self.By[item_idx]
        s = (y @ Sy.T + SBy[None, :]).astype(self.dtype)
        kind = self.final_act
        if kind[0] == 'softmax_logit':
            kind = ('softmax', 0.0, 0.0)
        yhat = final_act_fwd(*kind, s, np.ones(s.shape[1], dtype=bool)).astype(self.dtype)
        return yhat, newH


def ranks_from_scores(yhat, target_cols, mode='standard', tie=None):
    """evaluation.py:55-65 for the all-items case: yhat[M, I], target_cols[M].  mode 'tiebreaking' adds uniform * 1e-10 to
    every score in float32 (:55) and ranks as 'standard'; `tie` = (seed, step counter) of the Philox stream that stands in
    for the reference's MRG stream (only scores below ~2e-3 in magnitude are moved by it at all)."""
    if mode == 'tiebreaking':
        seed, ctr = tie
        u = philox.uniform_rows(yhat.shape[0], yhat.shape[1], seed, ctr, philox.STREAM_TIEBREAK)
        yhat = (yhat.astype(np.float32) + u * np.float32(1e-10)).astype(np.float32)
        mode = 'standard'
    t = yhat[np.arange(len(target_cols)), target_cols][:, None]
    gt = (yhat > t).sum(axis=1)
    if mode == 'standard':
        return gt + 1
    if mode == 'conservative':
        return (yhat >= t).sum(axis=1)
    if mode == 'median':
        return gt + 0.5 * ((yhat == t).sum(axis=1) - 1) + 1
    raise NotImplementedError(mode)
