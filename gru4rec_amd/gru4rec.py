"""GRU4Rec on MI355X: the reference's `GRU4Rec` class surface over hand-written gfx950 kernels.

Mirrors hidasib/GRU4Rec `gru4rec.py` (class GRU4Rec :27; __init__ :97-135; set_params :162-187;
fit :515-664; predict_next_batch :665-728; savemodel/loadmodel :742-781) so that `run.py -g
gru4rec_amd.gru4rec` / `evaluation.evaluate_gpu` keep working, but nothing here builds a Theano graph:
`fit` turns the session-parallel loop into an epoch plan (host scheduler, C++), uploads it once and
lets the device run whole epochs without a host round trip (libgru4rec_hip.so through ctypes).

There is deliberately no CPU fallback.
"""
import os
import pickle
import sys
import time
from collections import OrderedDict  # noqa: F401  (kept: parameter files use it)

import numpy as np
import pandas as pd

from . import _native, datatools, eventio
from .plan import build_rank_plan, pad_plan

_PLAIN_ACTS = ('linear', 'relu', 'tanh', 'softmax')


def _parse_act(name, allow_softmax):
    """'elu-0.5' -> (act id, p0, p1); unknown names raise NotImplementedError (gru4rec.py:144-161)."""
    if name in _PLAIN_ACTS:
        if name == 'softmax' and not allow_softmax:
            raise NotImplementedError
        return _native.ACT_IDS[name], 0.0, 0.0
    if name == 'softmax_logit' and allow_softmax:      # final activation only (gru4rec.py:149)
        return _native.ACT_IDS[name], 0.0, 0.0
    for prefix, npar in (('leaky-', 1), ('elu-', 1), ('selu-', 2)):
        if name.startswith(prefix):
            p = [float(x) for x in name.split('-')[1:]]
            if len(p) < npar:
                raise NotImplementedError
            return _native.ACT_IDS[prefix[:-1]], p[0], (p[1] if npar == 2 else 0.0)
    raise NotImplementedError


def _pad4(n):
    return (int(n) + 3) // 4 * 4


def _pad_cols(a, nblk, D, Dp):
    """(…, nblk * D) -> (…, nblk * Dp): every one of the nblk column blocks gets Dp - D zero columns."""
    if D == Dp:
        return a
    a = np.asarray(a)
    out = np.zeros(a.shape[:-1] + (nblk * Dp,), dtype=a.dtype)
    for b in range(nblk):
        out[..., b * Dp:b * Dp + D] = a[..., b * D:(b + 1) * D]
    return out


def _strip_cols(a, nblk, D, Dp):
    if D == Dp:
        return a
    return np.concatenate([a[..., b * Dp:b * Dp + D] for b in range(nblk)], axis=-1)


def _markers(*names):
    """Methods that only carry a name (a bound method pickles as getattr(obj, __name__))."""
    out = []
    for name in names:
        def f(self, *a, **k):
            raise NotImplementedError('symbolic Theano expression in the reference; computed by the HIP kernels here')
        f.__name__ = name
        f.__qualname__ = 'GRU4Rec.' + name
        out.append(f)
    return out


class GRU4Rec:
    """Same constructor arguments and defaults as the reference (gru4rec.py:97-101)."""

    accepts_categorical_items = True      # fit / evaluate_gpu take tables from eventio.read_events (run.py asks)

    def __init__(self, loss='bpr-max', final_act='linear', hidden_act='tanh', layers=[100],
                 n_epochs=10, batch_size=32, dropout_p_hidden=0.0, dropout_p_embed=0.0, learning_rate=0.1,
                 momentum=0.0, lmbd=0.0, embedding=0, n_sample=2048, sample_alpha=0.75, smoothing=0.0,
                 constrained_embedding=False, adapt='adagrad', adapt_params=[], grad_cap=0.0, bpreg=1.0, logq=0.0,
                 sigma=0.0, init_as_normal=False, train_random_order=False, time_sort=True,
                 session_key='SessionId', item_key='ItemId', time_key='Time'):
        self.layers = layers
        self.n_epochs = n_epochs
        self.batch_size = batch_size
        self.dropout_p_hidden = dropout_p_hidden
        self.dropout_p_embed = dropout_p_embed
        self.learning_rate = learning_rate
        self.adapt_params = adapt_params
        self.momentum = momentum
        self.sigma = sigma
        self.init_as_normal = init_as_normal
        self.session_key = session_key
        self.item_key = item_key
        self.time_key = time_key
        self.grad_cap = grad_cap
        self.bpreg = bpreg
        self.logq = logq
        self.train_random_order = train_random_order
        self.lmbd = lmbd
        self.embedding = self.layers[0] if embedding == 'layersize' else embedding
        self.constrained_embedding = constrained_embedding
        self.time_sort = time_sort
        self.adapt = adapt
        self.loss = loss
        self.set_loss_function(self.loss)
        self.final_act = final_act
        self.set_final_activation(self.final_act)
        self.hidden_act = hidden_act
        self.set_hidden_activation(self.hidden_act)
        self.n_sample = n_sample
        self.sample_alpha = sample_alpha
        self.smoothing = smoothing
        # ---- MI355X-path extras (not in the reference)
        self.seed = 12345            # Philox key (the reference's MRG default seed is 12345 as well)
        self.device = 0
        self.use_graph = True
        self.steps_per_call = 16384  # plan steps per C-ABI call (NaN check granularity)
        # multi-GPU runs: the GPU-local item tables are reconciled every `sync_every` steps and at the end of every epoch.  Measured
        # with virtual ranks (profiles/r03_virtual_ranks.json): reconciling only per epoch lets the replicas'
        # embedding spaces drift apart under the shared (all-reduced) GRU weights -- Recall@20 0.41 -> 0.15 at two ranks.  'auto':
        # 4 steps at two ranks (every 16: 0.20, every 4: 0.40 against 0.41 on one rank), 16 from three ranks on (four / eight ranks lose
        # 0.01 - 0.015 against every 4 and the exchange moves about the whole table each time); an integer is taken as given, 0 / None
        # reconciles at the end of the epoch only
        self.sync_every = 'auto'
        # multi-GPU runs, the other way to keep the replicas together: sparse_exact = True all-gathers every rank's per-occurrence
        # gradient rows of the gathered item rows EVERY step and every rank applies all of them in rank order with the reference's
        # duplicate semantics (gru4rec.py:335-340,407-431 over the concatenated occurrence list): the item tables never diverge, there
        # is nothing to reconcile and no sync_every to choose (SURVEY 8e option 3).  Costs an all-gather of R x D floats per rank and
        # step and needs nranks x (2 batch_size + n_sample) list entries in LDS: for small-catalogue shapes.
        # All ranks then draw the SAME negatives (one sample stream: the global batch shares its row of negatives, as the reference's
        # batch does, gru4rec.py:436-437).  True / 'reduce': the gradient rows of the shared negatives are SUMMED over the ranks (what
        # an all-reduce would give), the ranks' input / target occurrences are listed one rank behind the other, everything scaled to
        # the global batch -- the occurrence list of ONE batch of nranks x batch_size rows, updated as the reference updates its batch
        # EXCEPT that a row is scored against its own rank's batch_size in-batch negatives only (the hidden states live on their
        # ranks), not against all nranks x batch_size targets.  Kept for the A/B of DESIGN.md section 7: 'mean' (every rank's occurrences listed, an item's increment = the mean
        # over the ranks touching it) and 'sum' (every occurrence of every rank applied like a duplicate: diverges from four ranks on)
        self.sparse_exact = False
        # single GPU, Adagrad without momentum / lmbd: row updates whose item is not gathered again inside the current window of 16 steps
        # wait for ONE flush launch per window (g4r_config::defer_updates).  Results are bit-identical; the flush launch runs at ~60 % of
        # the HBM peak at BASELINE configs[2] -- and the step gets 2-5 % slower: off unless asked for
        self.defer_updates = False
        self._model = None
        self._dist = None
        self._cpu_store = False
        self.loss_history = []
        self.optimizer_state = None   # filled by savemodel(fname, optimizer_state=True); read by fit(resume=True)
        self.epochs_done = 0

    # ------------------------------------------------------------------ validation of names
    # The reference keeps its loss / activation as bound methods (`self.loss_function = self.bpr_max`, gru4rec.py:136-161)
    # and pickles them with the model (:742-756).  The same attribute names are kept here so that checkpoints travel both
    # ways (a pickle written by either implementation names `gru4rec.GRU4Rec` and these methods); the methods themselves
    # are markers: the arithmetic lives in the HIP kernels (k_loss_rows, act_fwd).
    linear, tanh, softmax, softmax_logit, softmax_neg, relu, sigmoid = _markers(
        'linear', 'tanh', 'softmax', 'softmax_logit', 'softmax_neg', 'relu', 'sigmoid')
    cross_entropy, cross_entropy_logits, bpr, bpr_max, top1, top1_max = _markers(
        'cross_entropy', 'cross_entropy_logits', 'bpr', 'bpr_max', 'top1', 'top1_max')

    class Selu:
        def __init__(self, lmbd, alpha):
            self.lmbd, self.alpha = lmbd, alpha

        def execute(self, X):
            raise NotImplementedError

    class Elu:
        def __init__(self, alpha):
            self.alpha = alpha

        def execute(self, X):
            raise NotImplementedError

    class LeakyReLU:
        def __init__(self, leak):
            self.leak = leak

        def execute(self, X):
            raise NotImplementedError

    _LOSS_METHODS = {'cross-entropy': 'cross_entropy', 'bpr': 'bpr', 'bpr-max': 'bpr_max', 'top1': 'top1',
                     'top1-max': 'top1_max', 'xe_logit': 'cross_entropy_logits'}

    def _act_callable(self, name):
        if name in ('linear', 'relu', 'tanh', 'softmax', 'softmax_logit'):
            return getattr(self, name)
        p = [float(x) for x in name.split('-')[1:]]
        if name.startswith('leaky-'):
            return self.LeakyReLU(p[0]).execute
        if name.startswith('elu-'):
            return self.Elu(p[0]).execute
        return self.Selu(*p).execute

    def set_loss_function(self, loss):
        if loss in _native.LOSS_IDS:         # gru4rec.py:136-143
            self._loss_id = _native.LOSS_IDS[loss]
            self.loss_function = getattr(self, self._LOSS_METHODS[loss])
        else:
            raise NotImplementedError

    def set_final_activation(self, final_act):
        self._final = _parse_act(final_act, True)
        self.final_activation = self._act_callable(final_act)

    def set_hidden_activation(self, hidden_act):
        self._hidden = _parse_act(hidden_act, False)
        self.hidden_activation = self._act_callable(hidden_act)

    def set_params(self, **kvargs):
        """String -> typed coercion against the current attribute type, as gru4rec.py:162-187."""
        kw = max(len(str(k)) for k in kvargs.keys())
        vw = max(len(str(v)) for v in kvargs.values())

        def show(k):
            val = getattr(self, k)
            print('SET   {}{}TO   {}{}(type: {})'.format(k, ' ' * (kw - len(k) + 3), val,
                                                          ' ' * (vw - len(str(val)) + 3), type(val)))
        for k, v in kvargs.items():
            if not hasattr(self, k) or k.startswith('_'):
                print('Unkown attribute: {}'.format(k))
                raise NotImplementedError
            cur = getattr(self, k)
            if isinstance(v, str):
                if k == 'adapt_params':
                    v = [float(x) for x in v.split('/')]
                elif isinstance(cur, list):
                    v = [int(x) for x in v.split('/')]
                elif isinstance(cur, bool):
                    if v in ('True', '1'):
                        v = True
                    elif v in ('False', '0'):
                        v = False
                    else:
                        print('Invalid value for boolean parameter: {}'.format(v))
                        raise NotImplementedError
            if k == 'embedding' and v == 'layersize':
                self.embedding = 'layersize'
            setattr(self, k, type(getattr(self, k))(v))
            if k == 'loss':
                self.set_loss_function(self.loss)
            if k == 'final_act':
                self.set_final_activation(self.final_act)
            if k == 'hidden_act':
                self.set_hidden_activation(self.hidden_act)
            show(k)
        if self.embedding == 'layersize':
            self.embedding = self.layers[0]
            show('embedding')
        self._check_limits()      # shapes the MI355X path does not serve are refused here, with the limit, not deep inside fit()

    # Shape limits of the HIP path (the reference has none of them; DESIGN.md section 5 "Limits" says where each comes from)
    MAX_WIDTH = 1024            # units of a GRU layer / of the item embedding: one gathered row = at most four 16-byte quads per lane
    LDS_BYTES = 156 * 1024      # what a workgroup of the step kernels may take of a CU's 160 KB

    def _check_limits(self):
        """NotImplementedError (the reference's way of refusing a configuration, gru4rec.py:143-177) naming the limit."""
        for D in self.layers:
            if _pad4(int(D)) > self.MAX_WIDTH:
                raise NotImplementedError('layers={}: the MI355X path serves GRU layers of up to {} units'.format(self.layers, self.MAX_WIDTH))
        if self.embedding and self.embedding != 'layersize' and _pad4(int(self.embedding)) > self.MAX_WIDTH:
            raise NotImplementedError('embedding={}: the MI355X path serves item embeddings of up to {} units'.format(self.embedding, self.MAX_WIDTH))
        if not self.constrained_embedding and not self.embedding and 3 * _pad4(int(self.layers[0])) > self.MAX_WIDTH:
            raise NotImplementedError('one-hot input (embedding=0, constrained_embedding=False) with layers[0]={}: the rows of Wx[0] are 3 * layers[0] wide '
                                      'and the MI355X path serves rows of up to {} floats (layers[0] <= {}); use constrained_embedding=True or '
                                      'embedding=<size> for wider first layers'.format(self.layers[0], self.MAX_WIDTH, self.MAX_WIDTH // 3 // 4 * 4))
        # one copy of a score row (k_loss_rows) and the step's occurrence list + partial rows (sparse update) live in LDS
        B, ns = int(self.batch_size), int(self.n_sample)
        ld = (B + ns + 15) // 16 * 16
        width = max([_pad4(int(self.layers[-1]))] + ([_pad4(int(self.embedding))] if (self.embedding and self.embedding != 'layersize') else [])
                    + ([3 * _pad4(int(self.layers[0]))] if (not self.constrained_embedding and not self.embedding) else []))
        rpad = ((2 * B + ns + 255) // 256) * 256 + 256
        need = max(4 * (ld + 288), 4 * rpad + 2112 + 32 * (width + 4))
        if need > self.LDS_BYTES:
            raise NotImplementedError('batch_size={} with n_sample={}: the MI355X path keeps a score row ({} columns) and the step\'s list of '
                                      '{} gathered rows in the 160 KB of LDS of a compute unit; that holds about 38,000 rows / columns '
                                      '(e.g. batch_size 512 with n_sample 36,000), fewer with rows wider than 512 units'.format(B, ns, B + ns, 2 * B + ns))

    # ------------------------------------------------------------------ weights (gru4rec.py:252-294)
    def _init_matrix(self, shape):
        sigma = self.sigma if self.sigma != 0 else np.sqrt(6.0 / (shape[0] + shape[1]))
        if self.init_as_normal:
            return np.asarray(np.random.randn(*shape) * sigma, dtype=np.float32)
        return np.asarray(np.random.rand(*shape) * sigma * 2 - sigma, dtype=np.float32)

    def _init_host_weights(self):
        """Same RNG draw order as the reference after np.random.seed(42): [E], per layer 3 Wx blocks, Wh,
        2 Wrz blocks, finally Wy -- so both implementations start from identical weights."""
        np.random.seed(42)
        L = self.layers
        if self.constrained_embedding:
            n_features = L[-1]
        elif self.embedding:
            self.E = self._init_matrix((self.n_items, self.embedding))
            n_features = self.embedding
        else:
            n_features = self.n_items
        self.Wx, self.Wh, self.Wrz, self.Bh, self.H = [], [], [], [], []
        for i, D in enumerate(L):
            n_in = L[i - 1] if i > 0 else n_features
            self.Wx.append(np.hstack([self._init_matrix((n_in, D)) for _ in range(3)]))
            self.Wh.append(self._init_matrix((D, D)))
            self.Wrz.append(np.hstack([self._init_matrix((D, D)) for _ in range(2)]))
            self.Bh.append(np.zeros(3 * D, dtype=np.float32))
            self.H.append(np.zeros((self.batch_size, D), dtype=np.float32))
        self.Wy = self._init_matrix((self.n_items, L[-1]))
        self.By = np.zeros((self.n_items, 1), dtype=np.float32)

    # ------------------------------------------------------------------ native model management
    def _check_supported(self):
        need = {'rmsprop': 1, 'adadelta': 1, 'adam': 2}.get(self.adapt, 0)
        if len(self.adapt_params) < need:
            raise IndexError('adapt={} needs {} value(s) in adapt_params'.format(self.adapt, need))     # the reference indexes adapt_params[0..1]
        if self.smoothing and self.loss not in ('cross-entropy', 'xe_logit'):
            raise NotImplementedError('smoothing is only defined for cross-entropy / xe_logit (gru4rec.py:226-235)')
        self._check_limits()

    def _create_model(self, sample_store, batch_size=None):
        self._check_supported()
        if self.adapt == 'adadelta' and self.learning_rate != 1.0:      # gru4rec.py:362-364
            print('Warn: learning_rate is not 1.0 while using adadelta. Setting learning_rate to 1.0')
            self.learning_rate = 1.0
        nranks = self._dist['nranks'] if self._dist else 1
        rank = self._dist['rank'] if self._dist else 0
        m = _native.Model(
            n_items=int(self.n_items), layers=[_pad4(D) for D in self.layers], batch_size=int(batch_size or self.batch_size),
            n_sample=int(self.n_sample), loss=self._loss_id,
            final_act=self._final[0], final_act_p0=self._final[1], final_act_p1=self._final[2],
            hidden_act=self._hidden[0], hidden_act_p0=self._hidden[1], hidden_act_p1=self._hidden[2],
            embed_mode=_native.EMBED_CONSTRAINED if self.constrained_embedding else (
                _native.EMBED_SEPARATE if self.embedding else _native.EMBED_ONEHOT),
            embedding=_pad4(self.embedding or 0), learning_rate=self.learning_rate, momentum=self.momentum,
            lmbd=self.lmbd, bpreg=self.bpreg, logq=self.logq, sample_alpha=self.sample_alpha, smoothing=float(self.smoothing),
            adapt=_native.ADAPT_IDS.get(self.adapt, _native.ADAPT_IDS[None]),      # any other value: plain SGD (gru4rec.py:392-399 fall through)
            adapt_p0=float(self.adapt_params[0]) if len(self.adapt_params) > 0 else 0.0,
            adapt_p1=float(self.adapt_params[1]) if len(self.adapt_params) > 1 else 0.0, grad_cap=float(self.grad_cap),
            dropout_p_hidden=self.dropout_p_hidden, dropout_p_embed=self.dropout_p_embed,
            # sample stream: every rank its own (GPU-local mode: more distinct negatives per global step), or -- exact-replica mode --
            # ONE stream for all ranks: the global batch then shares one row of negatives per step, as the reference's batch does
            # (gru4rec.py:436-437), and every sampled item is touched by all ranks, whose increments are averaged
            sample_store=int(sample_store), seed=int(self.seed) + (0 if self.sparse_exact else 7919 * rank), device=int(self.device),
            rank=rank, nranks=nranks, use_graph=1 if self.use_graph else 0,
            sparse_exact=({'sum': 1, 'mean': 2, 'reduce': 3}.get(self.sparse_exact, 3) if (self.sparse_exact and nranks > 1) else 0),
            defer_updates=1 if getattr(self, 'defer_updates', False) else 0)
        if getattr(self, 'defer_updates', False) and not m.get_debug('defer_stats', 4)[2]:
            import warnings
            warnings.warn('defer_updates is set but cannot apply to this model (it needs a single GPU, Adagrad without momentum, lmbd = 0, '
                          'no grad_cap): row updates are applied every step as usual')
        if self._dist and self._dist['unique_id'] is not None:
            m.comm_init(self._dist['unique_id'], nranks, rank)
            if nranks > 1:
                # every rank holds the communicator once this max-reduce returns: the rendezvous file has done its job and goes
                # now, not after fit() (a crash during training must not leave an id behind for the next run to read)
                from . import launch
                launch.barrier(m)
                launch.cleanup(rank)
            if os.environ.get('G4R_P2P') == '1' and nranks <= 8:
                # the switch next to the RCCL all-reduce: every rank reads its peers' gradients over xGMI itself (g4r_p2p_enable)
                m.p2p_enable()
        return m

    # ---- device layout: the library wants layer / embedding widths that are multiples of 4 (16-byte rows).  Other widths are
    # padded with zero columns / rows on the way to the device and stripped on the way back.  A padded unit has zero weights in
    # and out, so its activation is act(0) = 0, its state stays 0, and every gradient, accumulator and update that touches it
    # is exactly 0 (g = 0 gives a zero step under every `adapt`): the computation on the real units is unchanged.
    def _dev_spec(self, name, layer):
        """(rows, padded rows, column blocks, block width, padded block width) of a parameter array; rows = None for vectors."""
        base = name.split('_', 1)[1] if name.split('_', 1)[0] in ('acc', 'vel', 'acc2', 'cnt') else name
        L = self.layers
        D = L[layer] if base in ('Wx', 'Wh', 'Wrz', 'Bh', 'H') else L[-1]
        if base == 'Wx':
            if layer > 0:
                n_in = L[layer - 1]
            elif self.constrained_embedding:
                n_in = L[-1]
            elif self.embedding:
                n_in = self.embedding
            else:
                return self.n_items, self.n_items, 3, D, _pad4(D)      # one-hot input: Wx[0] is the (n_items, 3D) row table
            return n_in, _pad4(n_in), 3, D, _pad4(D)
        if base == 'Wh':
            return D, _pad4(D), 1, D, _pad4(D)
        if base == 'Wrz':
            return D, _pad4(D), 2, D, _pad4(D)
        if base == 'Bh':
            return None, None, 3, D, _pad4(D)
        if base == 'H':
            return None, None, 1, D, _pad4(D)      # rows = batch: taken from the array
        if base == 'Wy':
            return self.n_items, self.n_items, 1, D, _pad4(D)
        if base == 'E':
            return self.n_items, self.n_items, 1, self.embedding, _pad4(self.embedding)
        if base == 'By':
            return None, None, 1, 1, 1
        raise KeyError(name)

    def _dev_put(self, m, name, arr, layer=0):
        rows, rows_p, nblk, D, Dp = self._dev_spec(name, layer)
        a = _pad_cols(np.asarray(arr, dtype=np.float32), nblk, D, Dp) if name.split('_')[-1] != 'By' else np.asarray(arr, dtype=np.float32).reshape(-1)
        if rows is not None and rows_p != rows:
            a = np.concatenate([a, np.zeros((rows_p - rows, a.shape[1]), dtype=np.float32)])
        m.set_param(name, a, layer)

    def _dev_get(self, m, name, shape, layer=0):
        rows, rows_p, nblk, D, Dp = self._dev_spec(name, layer)
        if name.split('_')[-1] == 'By':
            return m.get_param(name, (self.n_items,), layer).reshape(shape)
        if rows is None:
            pshape = tuple(shape[:-1]) + (nblk * Dp,)
        else:
            pshape = (rows_p, nblk * Dp)
        a = _strip_cols(m.get_param(name, pshape, layer), nblk, D, Dp)
        if rows is not None and rows_p != rows:
            a = a[:rows]
        return np.ascontiguousarray(a).reshape(shape)

    def _upload_weights(self, m):
        for i in range(len(self.layers)):
            self._dev_put(m, 'Wx', self.Wx[i], i)
            self._dev_put(m, 'Wh', self.Wh[i], i)
            self._dev_put(m, 'Wrz', self.Wrz[i], i)
            self._dev_put(m, 'Bh', self.Bh[i], i)
        self._dev_put(m, 'Wy', self.Wy)
        self._dev_put(m, 'By', self.By.reshape(-1))
        if not self.constrained_embedding and self.embedding:
            self._dev_put(m, 'E', self.E)

    def _download_weights(self):
        m = self._model
        L = self.layers
        for i, D in enumerate(L):
            n_in = self.Wx[i].shape[0]
            self.Wx[i] = self._dev_get(m, 'Wx', (n_in, 3 * D), i)
            self.Wh[i] = self._dev_get(m, 'Wh', (D, D), i)
            self.Wrz[i] = self._dev_get(m, 'Wrz', (D, 2 * D), i)
            self.Bh[i] = self._dev_get(m, 'Bh', (3 * D,), i)
            self.H[i] = self._dev_get(m, 'H', (m.cfg.batch_size, D), i)
        self.Wy = self._dev_get(m, 'Wy', (self.n_items, L[-1]))
        self.By = self._dev_get(m, 'By', (self.n_items,)).reshape(-1, 1)
        if not self.constrained_embedding and self.embedding:
            self.E = self._dev_get(m, 'E', (self.n_items, self.embedding))

    # ---- optimizer state ("plus new optimizer-state save"; the reference's pickles hold the weights only)
    def _opt_tables(self):
        """(name, layer, shape) of every optimizer-state array the device keeps for this configuration."""
        pre = ['acc_']
        if self.momentum > 0:
            pre.append('vel_')
        if self.adapt in ('adadelta', 'adam'):
            pre.append('acc2_')
        if self.adapt == 'adam':
            pre.append('cnt_')
        if self.adapt not in ('adagrad', 'rmsprop', 'adadelta', 'adam'):
            pre = [p for p in pre if p == 'vel_']      # plain SGD keeps no statistics
        out = []
        L = self.layers
        for p in pre:
            for i, D in enumerate(L):
                out += [(p + 'Wx', i, (self.Wx[i].shape[0], 3 * D)), (p + 'Wh', i, (D, D)), (p + 'Wrz', i, (D, 2 * D)), (p + 'Bh', i, (3 * D,))]
            out += [(p + 'Wy', 0, (self.n_items, L[-1])), (p + 'By', 0, (self.n_items,))]
            if not self.constrained_embedding and self.embedding:
                out.append((p + 'E', 0, (self.n_items, self.embedding)))
        return out

    def _download_optimizer_state(self):
        m = self._model
        # np_random_state: the session order of train_random_order comes from NumPy's global stream (np.random.permutation per
        # epoch, gru4rec.py:593), seeded by the weight initialisation; a resumed run has to continue THAT stream
        st = {'arrays': {}, 'global_step': m.global_step(), 'refills': m.refills(), 'np_random_state': np.random.get_state(),
              'config': self._opt_config()}
        for name, layer, shape in self._opt_tables():
            st['arrays'][(name, layer)] = self._dev_get(m, name, shape, layer)
        return st

    def _opt_config(self):
        """What the saved optimizer state is a function of: resuming under another value of any of these would either not find its
        arrays or silently drop some (e.g. the velocities when momentum goes to 0)."""
        return dict(adapt=self.adapt, adapt_params=[float(x) for x in self.adapt_params], momentum=float(self.momentum),
                    layers=[int(x) for x in self.layers], embedding=int(self.embedding or 0),
                    constrained_embedding=bool(self.constrained_embedding), batch_size=int(self.batch_size), n_sample=int(self.n_sample))

    def _upload_optimizer_state(self, m, st):
        for name, layer, shape in self._opt_tables():
            self._dev_put(m, name, st['arrays'][(name, layer)], layer)
        m.set_step_counters(st['global_step'], st['refills'])

    def set_distributed(self, rank, nranks, unique_id):
        """One process per GPU: sessions are sharded round-robin over ranks, dense GRU gradients are
        all-reduced with RCCL every step, embedding rows stay GPU-local.
        unique_id = None: a virtual rank (gru4rec_amd/virtual_ranks.py: several handles of ONE process stand in for the ranks; no
        communicator is created and the caller steps the handles together)."""
        # (a one-rank layout with an id: the N > 1 data path on a one-rank communicator, G4R_FORCE_STAGED=1 -- tests)
        self._dist = dict(rank=int(rank), nranks=int(nranks), unique_id=unique_id) if (nranks > 1 or unique_id is not None) else None

    # ------------------------------------------------------------------ training (gru4rec.py:515-664)
    def prepare(self, data, sample_store=10000000, store_type='gpu', resume=False):
        """Everything fit() does before its epoch loop: item map, sort, offsets, weights, popularity tables,
        device model.  Split out so that benchmarks can time the epoch loop alone (the reference's own
        mb/s excludes these as well).  resume=True keeps the weights on this object (a loaded checkpoint) instead of
        initialising them, and restores the optimizer state / step counters saved next to them."""
        if store_type not in ('gpu', 'cpu'):
            print('Invalid store type {}'.format(store_type))
            raise NotImplementedError
        if resume and store_type == 'cpu' and self.n_sample:
            # before anything is built (the reference validates store_type first as well, gru4rec.py:546-555)
            raise NotImplementedError('resume=True with store_type="cpu": the host sampler interleaves its draws with the session '
                                      'order on NumPy\'s global random stream; only the device store (store_type="gpu") resumes')
        self.predict = None
        self.error_during_train = False
        item_col = data[self.item_key]
        if eventio.is_categorical(item_col):
            # table from eventio.read_events: the category codes already are the indices (no hash join over the events)
            itemids, item_idx = eventio.first_appearance_index(item_col)
            itemidmap = pd.Series(data=np.arange(len(itemids)), index=itemids, name='ItemIdx')
            data['ItemIdx'] = item_idx
        else:
            itemids = item_col.unique()
            itemidmap = pd.Series(data=np.arange(len(itemids)), index=itemids, name='ItemIdx')
            data['ItemIdx'] = itemidmap[item_col.values].values
        if resume:
            if not hasattr(self, 'Wy') or getattr(self, 'optimizer_state', None) is None:
                raise ValueError('resume=True needs a model saved with savemodel(fname, optimizer_state=True)')
            if len(itemids) != self.n_items or not np.array_equal(np.asarray(itemidmap.index), np.asarray(self.itemidmap.index)):
                raise ValueError('resume=True: the training data does not produce the item map of the checkpoint')
            saved_cfg = self.optimizer_state.get('config')
            if saved_cfg is not None and saved_cfg != self._opt_config():
                diff = sorted(k for k in saved_cfg if saved_cfg[k] != self._opt_config().get(k))
                raise ValueError('resume=True: %s differ(s) from the checkpoint (%s), its optimizer state does not apply' % (
                    ', '.join(diff), ', '.join('%s=%r' % (k, saved_cfg[k]) for k in diff)))
            if self.train_random_order and self.optimizer_state.get('np_random_state') is None:
                raise ValueError('resume=True with train_random_order: the checkpoint does not hold the random stream of the session order')
        self.n_items = len(itemids)
        self.itemidmap = itemidmap
        datatools.sort_if_needed(data, [self.session_key, self.time_key])
        self._offsets = datatools.compute_offset(data, self.session_key)
        if not resume:
            self._init_host_weights()
            self.optimizer_state = None
            self.epochs_done = 0
        # events per item in itemidmap order: data.groupby(item_key).size()[itemidmap.index] of gru4rec.py:540-541
        support = np.bincount(data['ItemIdx'].values, minlength=self.n_items)
        if self._model is not None:
            self._model.close()
        self._model = self._create_model(sample_store)
        m = self._model
        self._upload_weights(m)
        self._cpu_store = bool(store_type == 'cpu' and self.n_sample)
        if self.n_sample and m.sample_store_rows() <= 1:
            print('No example store was used')      # negatives are then drawn anew for every step (gru4rec.py:548-550,614-615)
        lq_t = lq_s = None
        if self.logq:
            p0 = support.astype(np.float32)
            lq_t = np.log(p0)
            lq_s = np.log(p0 ** np.float32(self.sample_alpha))
        pop = support.astype(np.float64) ** self.sample_alpha
        pop = pop.cumsum() / pop.sum()
        pop[-1] = 1
        self._pop64 = pop
        if self._cpu_store:
            # store_type='cpu' (gru4rec.py:551-554): the reference's own host sampler, on NumPy's global stream right behind the
            # weight initialisation -- the negatives are the reference's, draw for draw
            m.set_sample_store(self._cpu_samples(m.sample_store_rows()))
        m.set_popularity(pop.astype(np.float32), lq_t, lq_s)
        if self.n_sample and m.sample_store_rows() > 1:
            print('Created sample store with {} batches of samples (type={})'.format(m.sample_store_rows(), 'CPU' if self._cpu_store else 'GPU'))
        if resume:
            self._upload_optimizer_state(m, self.optimizer_state)
            if self.optimizer_state.get('np_random_state') is not None:
                np.random.set_state(self.optimizer_state['np_random_state'])      # continue the stream of np.random.permutation (:593)
        if self.time_sort:
            # data is ordered by (session, time): a session's first row holds its minimum time (the groupby().min() of
            # gru4rec.py:585-586, sessions in ascending id order)
            self._base_order = np.argsort(data[self.time_key].values[self._offsets[:-1]])
        else:
            self._base_order = np.arange(len(self._offsets) - 1)
        self._data_items = data.ItemIdx.values.astype(np.int32)
        self._plan_key = None
        if not resume:
            self.loss_history = []
        self.step_costs = []          # per-epoch arrays of the per-mini-batch cost (gru4rec.py:623)

    def _cpu_samples(self, length):
        """generate_neg_samples, gru4rec.py:507-514: searchsorted (side='left') in the float64 cumulative table, or a uniform
        choice when sample_alpha == 0; `length` rows of n_sample."""
        if self.sample_alpha:
            sample = np.searchsorted(self._pop64, np.random.rand(self.n_sample * length))
        else:
            sample = np.random.choice(self.n_items, size=self.n_sample * length)
        return sample.reshape(length, self.n_sample).astype(np.int32)

    def _epoch_plan(self):
        n_sessions = len(self._offsets) - 1
        order_all = np.random.permutation(n_sessions) if self.train_random_order else self._base_order
        key = None if self.train_random_order else 'static'
        if key is not None and self._plan_key == key:
            return self._plan
        if self._dist:
            plan = build_rank_plan(self._offsets, order_all, self._data_items, self.batch_size, self.n_sample,
                                   self._dist['rank'], self._dist['nranks'])
            # every rank must issue the same number of all-reduces: the shorter plans are padded with M = 0 steps
            plan = pad_plan(plan, self._model.comm_max(plan['T']))
        else:
            plan = build_rank_plan(self._offsets, order_all, self._data_items, self.batch_size, self.n_sample)
        self._model.set_plan(plan)
        self._plan = plan
        self._plan_key = key
        return plan

    def sync_steps(self, nranks=None):
        """Steps between two reconciliations of the GPU-local item tables for `nranks` ranks (default: this object's layout): the
        `sync_every` attribute, 'auto' resolved as documented there; 0 = only at the end of an epoch."""
        if nranks is None:
            nranks = self._dist['nranks'] if self._dist else 1
        if self.sparse_exact:
            return 0      # exact replicas: nothing to reconcile
        k = self.sync_every
        if k == 'auto':
            # catalogues too large for the on-stream dense reconciliation (item tables > 64 MB per group: the packed-parts exchange moves
            # the rows touched since the last call): every 64 steps.  Measured at a configs[3]-like shape (1 M items, layers [256],
            # B = 512, 8192 negatives, 8 virtual ranks; profiles/r04_virtual_ranks_large.json): Recall@20 0.092 / 0.094 / 0.093 / 0.091 at
            # every 4 / 16 / 64 steps / epoch end -- the ranks' rows rarely collide on a catalogue of that size, what is lost against
            # one rank (0.385) is the 8 x larger global batch (one rank at B = 4096: 0.098) -- while a reconciliation moves 0.20 / 0.55 /
            # 1.21 / 1.57 M rows: 64 keeps the exchange at ~19 K rows per step.
            width = max(_pad4(int(self.layers[-1])), 1)
            if getattr(self, 'n_items', 0) and int(self.n_items) * (2 * width + 2 + 1) * 4 > 64 * 1024 * 1024:
                return 64
            return 4 if nranks == 2 else 16
        if not k:
            return 0
        k = int(k)
        if k < 0:
            raise ValueError('sync_every must be a number of steps, 0 / None or "auto"')
        return k

    def run_epoch(self, epoch, max_steps=None):
        """One pass of the reference's epoch body (gru4rec.py:587-661).  Returns (costs, M per step) or None on NaN."""
        m = self._model
        t0 = time.time()
        plan = self._epoch_plan()
        m.reset_hidden()
        T = plan['T'] if max_steps is None else min(plan['T'], max_steps)
        costs = np.empty(T, dtype=np.float32)
        done = 0
        since_sync = 0
        # small item tables are reconciled inside the library (every sync_every steps, between two steps, on the stream); otherwise
        # the calls are cut at those points and g4r_comm_sync_sparse runs in between
        sync_k = self.sync_steps()
        host_sync = bool(self._dist and sync_k) and not m.set_sync_every(sync_k)
        while done < T:
            n = min(self.steps_per_call, T - done)
            if host_sync:
                n = min(n, sync_k - since_sync)      # every rank cuts at the same steps: plans have one common length
            if self._cpu_store:
                # host sample store: the row pointer is the global step modulo the store length; a new store is drawn when it
                # wraps (gru4rec.py:609-613), so no device call runs across that point
                g, gl = m.global_step(), m.sample_store_rows()
                if g > 0 and g % gl == 0:
                    m.set_sample_store(self._cpu_samples(gl))
                n = min(n, gl - g % gl)
            m.train_steps(done, n)
            c = m.get_losses(done, n)
            costs[done:done + n] = c
            bad = bool(np.isnan(c).any())
            if self._dist:
                bad = bool(m.comm_max(int(bad)))      # all ranks leave together: a lone return would hang the others' all-reduce
            if bad:
                print(str(epoch) + ': NaN error!')
                self.error_during_train = True
                return None
            done += n
            since_sync += n
            if host_sync and since_sync >= sync_k and done < T:
                m.comm_sync_sparse()
                since_sync = 0
        cc = plan['M'][:T]
        avgc = np.sum(costs * cc) / max(np.sum(cc), 1)
        bad = bool(np.isnan(avgc))
        if self._dist:
            bad = bool(m.comm_max(int(bad)))      # collective like the per-chunk exit: a lone return would hang the other ranks
        if bad:
            print('Epoch {}: NaN error!'.format(str(epoch)))
            self.error_during_train = True
            return None
        dt = time.time() - t0
        print('Epoch{} --> loss: {:.6f} \t({:.2f}s) \t[{:.2f} mb/s | {:.0f} e/s]'.format(epoch + 1, avgc, dt, T / dt, np.sum(cc) / dt))
        self.loss_history.append(float(avgc))
        self.step_costs.append(costs)
        self.last_epoch_stats = dict(steps=int(T), events=int(np.sum(cc)), seconds=dt, loss=float(avgc))
        return costs, cc

    def fit(self, data, sample_store=10000000, store_type='gpu', resume=False):
        """Trains the network; same contract as the reference's fit (gru4rec.py:515-664): mutates `data`
        (adds ItemIdx, may sort in place), sets n_items / itemidmap / error_during_train.
        resume=True (not in the reference): continue a model loaded from savemodel(fname, optimizer_state=True) on the same
        training data from the epoch it stopped at, up to n_epochs -- bit-identical to the uninterrupted run."""
        self.prepare(data, sample_store=sample_store, store_type=store_type, resume=resume)
        for epoch in range(self.epochs_done if resume else 0, self.n_epochs):
            if self.run_epoch(epoch) is None:
                return
            if self._dist:
                self._model.comm_sync_sparse()
            self.epochs_done = epoch + 1
        self._download_weights()

    def close(self):
        """Releases the device model (parameters stay on the host object); the next predict / fit creates a new one."""
        if self._model is not None:
            self._model.close()
            self._model = None
        self.predict = None

    # ------------------------------------------------------------------ prediction (gru4rec.py:665-728)
    def _ensure_model(self):
        if self._model is None:
            self._model = self._create_model(0)
            self._upload_weights(self._model)
        return self._model

    def predict_next_batch(self, session_ids, input_item_ids, predict_for_item_ids=None, batch=100):
        """Scores for the next item of every session in the batch.  Rows: items, columns: batch events."""
        if self.error_during_train:
            raise Exception
        m = self._ensure_model()
        if self.predict is None or self.predict_batch != batch:
            self.predict_batch = batch
            m.predict_begin(batch)
            self.current_session = np.ones(batch) * -1
            self.predict = True
        session_ids = np.asarray(session_ids)
        changed = session_ids != self.current_session
        if changed.any():
            m.predict_hidden(zero_mask=changed.astype(np.uint8))
            self.current_session = session_ids.copy()
        in_idxs = self.itemidmap[input_item_ids].values
        if predict_for_item_ids is not None:
            iidx = self.itemidmap[predict_for_item_ids].values
            preds = m.predict_step(in_idxs, iidx).T
            return pd.DataFrame(data=preds, index=predict_for_item_ids)
        preds = m.predict_step(in_idxs).T
        return pd.DataFrame(data=preds, index=self.itemidmap.index)

    def symbolic_predict(self, X, Y, M, items, batch_size):
        raise NotImplementedError('symbolic_predict builds a Theano graph (gru4rec.py:729-741); the MI355X path '
                                  'exposes the same computation through gru4rec_amd.evaluation.evaluate_gpu')

    # ------------------------------------------------------------------ (de)serialisation (gru4rec.py:742-781)
    _EXTRAS = dict(seed=12345, device=0, use_graph=True, steps_per_call=16384, sync_every='auto', sparse_exact=False, defer_updates=False)     # attributes the reference does not have

    def __getstate__(self):
        st = dict(self.__dict__)
        for k in ('_model', '_plan', '_plan_key', '_data_items', '_offsets', '_base_order', '_dist', '_loss_id', '_final', '_hidden', '_pop64', '_cpu_store'):
            st.pop(k, None)
        st['predict'] = None
        return st

    def __setstate__(self, st):
        """Accepts pickles of this class and of the reference's (whose state lacks the MI355X extras)."""
        self.__dict__.update(st)
        for k, v in self._EXTRAS.items():
            self.__dict__.setdefault(k, v)
        self.__dict__.setdefault('loss_history', [])
        self.__dict__.setdefault('optimizer_state', None)
        self.__dict__.setdefault('epochs_done', 0)
        self.__dict__.setdefault('error_during_train', False)
        self.set_loss_function(self.loss)
        self.set_final_activation(self.final_act)
        self.set_hidden_activation(self.hidden_act)
        if hasattr(self, 'By') and self.By is not None:
            self.By = np.asarray(self.By, dtype=np.float32).reshape(-1, 1)
        self._model = None
        self._dist = None
        self.predict = None

    def savemodel(self, fname, optimizer_state=False):
        """Pickle with the reference's attribute names / array layouts (Wx[i] (in,3D)=[cand|r|z], Wrz[i] (D,2D)=[r|z],
        Wh[i], Bh[i], H[i], Wy (n_items,D), By (n_items,1), E, itemidmap).  optimizer_state=True adds, under attributes the
        reference's loadmodel never looks at (`optimizer_state`, `epochs_done`), what fit(resume=True) needs to continue: the
        accumulator / velocity arrays, the global step and the sample-store refill count.  The file stays loadable by the
        reference (gru4rec.py:768-781 touches the weight attributes only)."""
        if self._model is not None and not self.error_during_train:
            self._download_weights()
        keep = getattr(self, 'optimizer_state', None)
        if optimizer_state:
            if self._model is None:
                raise ValueError('optimizer_state=True needs the trained device model (call savemodel before close())')
            self.optimizer_state = self._download_optimizer_state()
        else:
            self.optimizer_state = None
        try:
            with open(fname, 'wb') as f:
                pickle.dump(self, f)
        finally:
            if not optimizer_state:
                self.optimizer_state = keep

    @classmethod
    def loadmodel(cls, fname):
        """gru4rec.py:768-781.  Reads checkpoints written by this class and by the reference's `savemodel` (both name the
        class `gru4rec.GRU4Rec`; the reference's arrays are plain NumPy in the pickle, :744-756)."""
        class _Opaque:
            """Stand-in for Theano / pygpu objects inside a reference pickle (sample store, RNG state, compiled
            functions left on the object after fit): they are rebuilt by fit() / predict here, never read."""
            def __init__(self, *a, **k):
                pass

            def __setstate__(self, st):
                pass

        class _Unpickler(pickle.Unpickler):
            def find_class(self, module, name):
                if module.split('.')[0] in ('theano', 'pygpu'):
                    return _Opaque
                if module in ('gru4rec', 'gru4rec_amd.gru4rec'):
                    obj = sys.modules[cls.__module__] if cls.__module__ in sys.modules else None
                    target = cls
                    parts = name.split('.')
                    if parts[0] == 'GRU4Rec':
                        for part in parts[1:]:
                            target = getattr(target, part)
                        return target
                    if obj is not None and hasattr(obj, name):
                        return getattr(obj, name)
                return super().find_class(module, name)
        with open(fname, 'rb') as f:
            gru = _Unpickler(f).load()
        for k in [k for k, v in gru.__dict__.items() if isinstance(v, _Opaque)]:
            del gru.__dict__[k]
        gru._model = None
        gru.predict = None
        return gru


# Checkpoints name the class `gru4rec.GRU4Rec`, exactly like the reference's (gru4rec.py:755), so they load on either side;
# `gru4rec.py` at the repository root re-exports this class under that module name.
GRU4Rec.__module__ = 'gru4rec'
GRU4Rec.__qualname__ = 'GRU4Rec'
for _c in (GRU4Rec.Selu, GRU4Rec.Elu, GRU4Rec.LeakyReLU):
    _c.__module__ = 'gru4rec'
sys.modules.setdefault('gru4rec', sys.modules[__name__])     # so pickle can resolve the name without the root shim
