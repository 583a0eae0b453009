"""Minimal stand-in for Theano 1.0.5 -- TEST INFRASTRUCTURE, used only by oracle/make_golden.py.

The reference (hidasib/GRU4Rec) states its whole hot path as a Theano graph; Theano / libgpuarray are
third-party, un-vendored, unpinned ("1.0.5 or newer", README.md:45) and not installable here.  This shim
restates the published semantics of exactly the API subset that `/root/reference/gru4rec.py` and
`gpu_ops.py` call, so that the reference's OWN source text can be imported and executed to generate
golden vectors:

  * lazy symbolic expressions (`Var`), evaluated with torch (CPU, float32 = floatX) at call time
  * `theano.shared`, `theano.function(inputs, outputs, updates=...)` with simultaneous update semantics
  * `T.grad` = reverse-mode differentiation of the scalar cost w.r.t. shared variables or intermediate
    expressions (torch.autograd on the same graph)
  * `set_subtensor` (duplicate indices: last write wins, NumPy order) / `inc_subtensor` (duplicates accumulate)
  * random streams are routed to a hook so the caller decides which uniforms / masks are drawn

Nothing in the product imports this.
"""
import numpy as np
import torch


class _Config:
    floatX = 'float32'


config = _Config()
_TORCH_DT = {'float32': torch.float32, 'float64': torch.float64, 'int32': torch.int64, 'int64': torch.int64,
             'int8': torch.int64, 'bool': torch.bool}
_serial = [0]


def _as_var(x):
    if isinstance(x, Var):
        return x
    return Var('const', [], value=x, ndim=np.ndim(x))


class ShapeTuple:
    def __init__(self, var):
        self.var = var

    def __iter__(self):
        for i in range(self.var.ndim):
            yield Var('shape_elem', [self.var], axis=i, ndim=0)

    def __getitem__(self, i):
        return Var('shape_elem', [self.var], axis=i, ndim=0)

    def __len__(self):
        return self.var.ndim


class Var:
    """A node of the symbolic graph."""

    def __init__(self, op, inputs, ndim=None, name=None, **attrs):
        self.op = op
        self.inputs = [_as_var(i) if not isinstance(i, (Var, type(None))) else i for i in inputs]
        self.attrs = attrs
        self.ndim = ndim
        self.name = name
        _serial[0] += 1
        self.serial = _serial[0]

    # ---- arithmetic
    def _bin(self, other, fn, rev=False):
        o = _as_var(other)
        a, b = (o, self) if rev else (self, o)
        return Var('elem2', [a, b], fn=fn, ndim=max(a.ndim or 0, b.ndim or 0))

    def __add__(self, o): return self._bin(o, 'add')
    def __radd__(self, o): return self._bin(o, 'add', True)
    def __sub__(self, o): return self._bin(o, 'sub')
    def __rsub__(self, o): return self._bin(o, 'sub', True)
    def __mul__(self, o): return self._bin(o, 'mul')
    def __rmul__(self, o): return self._bin(o, 'mul', True)
    def __truediv__(self, o): return self._bin(o, 'div')
    def __rtruediv__(self, o): return self._bin(o, 'div', True)
    def __floordiv__(self, o): return self._bin(o, 'floordiv')
    def __pow__(self, o): return self._bin(o, 'pow')
    def __rpow__(self, o): return self._bin(o, 'pow', True)
    def __neg__(self): return Var('elem1', [self], fn='neg', ndim=self.ndim)
    def __gt__(self, o): return self._bin(o, 'gt')
    def __ge__(self, o): return self._bin(o, 'ge')
    def __lt__(self, o): return self._bin(o, 'lt')
    def __le__(self, o): return self._bin(o, 'le')
    def __eq__(self, o): return self._bin(o, 'eq')
    __hash__ = object.__hash__

    # ---- structure
    @property
    def shape(self):
        return ShapeTuple(self)

    @property
    def T(self):
        return Var('transpose', [self], ndim=self.ndim)

    @property
    def dtype(self):
        return self.attrs.get('dtype', config.floatX)

    def max(self, axis=None, keepdims=False):
        return Var('reduce', [self], fn='max', axis=axis, keepdims=keepdims,
                   ndim=(self.ndim if keepdims else (0 if axis is None else self.ndim - 1)))

    def sum(self, axis=None, keepdims=False):
        return Var('reduce', [self], fn='sum', axis=axis, keepdims=keepdims,
                   ndim=(self.ndim if keepdims else (0 if axis is None else self.ndim - 1)))

    def mean(self, axis=None, keepdims=False):
        return Var('reduce', [self], fn='mean', axis=axis, keepdims=keepdims,
                   ndim=(self.ndim if keepdims else (0 if axis is None else self.ndim - 1)))

    def flatten(self):
        return Var('flatten', [self], ndim=1)

    def reshape(self, shape):
        return Var('reshape', [self] + [_as_var(s) for s in shape], ndim=len(shape))

    def __getitem__(self, idx):
        if not isinstance(idx, tuple):
            idx = (idx,)
        spec, ins, nd = [], [self], self.ndim
        for it in idx:
            if isinstance(it, slice):
                parts = []
                for p in (it.start, it.stop, it.step):
                    if p is None:
                        parts.append(None)
                    else:
                        ins.append(_as_var(p))
                        parts.append(len(ins) - 1)
                spec.append(('slice', parts))
            else:
                v = _as_var(it)
                ins.append(v)
                spec.append(('index', len(ins) - 1))
                if (v.ndim or 0) == 0:
                    nd -= 1
        return Var('subtensor', ins, spec=spec, ndim=nd)


class SharedVariable(Var):
    def __init__(self, value, name=None):
        value = np.asarray(value)
        super().__init__('shared', [], ndim=value.ndim, name=name, dtype=str(value.dtype))
        self.value = value

    def get_value(self, borrow=False):
        return self.value if borrow else self.value.copy()

    def set_value(self, v, borrow=False):
        self.value = np.asarray(v)


def shared(value, borrow=False, name=None):
    return SharedVariable(value, name=name)


# ---------------------------------------------------------------------------------------------- evaluation
RNG_HOOK = [None]    # callable(kind, node_serial_rank, call_count, shape, attrs) -> ndarray
CALL_LOG = []        # (function id, outputs) of every compiled-function call, for the golden generator
_rng_nodes = []      # creation order of random nodes


def _to_t(x, float_dt=torch.float32):
    if isinstance(x, torch.Tensor):
        return x
    a = np.asarray(x)
    if a.dtype.kind == 'f':
        return torch.from_numpy(np.array(a, dtype=np.float32, order='C')).to(float_dt)
    if a.dtype.kind == 'b':
        return torch.from_numpy(np.array(a, order='C'))
    return torch.from_numpy(np.array(a, dtype=np.int64, order='C'))


class _Ctx:
    def __init__(self, env, call_count, rng_cache):
        self.env = env
        self.memo = {}
        self.call_count = call_count
        self.rng_cache = rng_cache
        self.leaf = {}


def _ev(v, c):
    if v.serial in c.leaf:
        return c.leaf[v.serial]
    if v.serial in c.memo:
        return c.memo[v.serial]
    r = _ev1(v, c)
    c.memo[v.serial] = r
    return r


def _idx_of(v, c):
    t = _ev(v, c)
    if isinstance(t, torch.Tensor):
        return t.long() if t.dtype not in (torch.bool,) else t
    return t


def _ev1(v, c):
    op = v.op
    I = v.inputs
    if op == 'const':
        return _to_t(v.attrs['value'])
    if op == 'input':
        return c.env[v.serial]
    if op == 'shared':
        return _to_t(v.value)
    if op == 'shape_elem':
        return torch.tensor(_ev(I[0], c).shape[v.attrs['axis']], dtype=torch.int64)
    if op == 'elem2':
        a, b = _ev(I[0], c), _ev(I[1], c)
        fn = v.attrs['fn']
        if fn == 'add': return a + b
        if fn == 'sub': return a - b
        if fn == 'mul': return a * b
        if fn == 'div':
            if not a.is_floating_point() and not b.is_floating_point():
                return a.float() / b.float()
            return a / b
        if fn == 'floordiv': return torch.div(a, b, rounding_mode='floor')
        if fn == 'pow': return a ** b
        if fn == 'gt': return a > b
        if fn == 'ge': return a >= b
        if fn == 'lt': return a < b
        if fn == 'le': return a <= b
        if fn == 'eq': return a == b
        if fn == 'maximum': return torch.maximum(a, b if isinstance(b, torch.Tensor) else torch.tensor(b))
        raise NotImplementedError(fn)
    if op == 'elem1':
        a = _ev(I[0], c)
        fn = v.attrs['fn']
        if fn == 'neg': return -a
        if fn == 'exp': return torch.exp(a)
        if fn == 'log': return torch.log(a)
        if fn == 'sqrt': return torch.sqrt(a)
        if fn == 'tanh': return torch.tanh(a)
        if fn == 'sigmoid': return torch.sigmoid(a)
        if fn == 'cast':
            return a.to(_TORCH_DT[v.attrs['to']])
        if fn == 'zeros_like': return torch.zeros_like(a, dtype=_TORCH_DT.get(v.attrs.get('to'), a.dtype))
        if fn == 'ones_like': return torch.ones_like(a, dtype=_TORCH_DT.get(v.attrs.get('to'), a.dtype))
        raise NotImplementedError(fn)
    if op == 'switch':
        cond, a, b = _ev(I[0], c), _ev(I[1], c), _ev(I[2], c)
        cond = cond.bool() if cond.dtype != torch.bool else cond
        if not isinstance(a, torch.Tensor) or a.ndim == 0:
            a = torch.as_tensor(a, dtype=b.dtype if isinstance(b, torch.Tensor) else torch.float32)
        if not isinstance(b, torch.Tensor) or b.ndim == 0:
            b = torch.as_tensor(b, dtype=a.dtype)
        return torch.where(cond, a.to(torch.result_type(a, b)), b.to(torch.result_type(a, b)))
    if op == 'dot':
        return _ev(I[0], c) @ _ev(I[1], c)
    if op == 'transpose':
        a = _ev(I[0], c)
        return a.t() if a.ndim == 2 else a
    if op == 'reduce':
        a = _ev(I[0], c)
        fn, axis, kd = v.attrs['fn'], v.attrs['axis'], v.attrs['keepdims']
        if a.dtype == torch.bool:
            a = a.long()
        if fn == 'sum':
            return a.sum() if axis is None else a.sum(dim=axis, keepdim=kd)
        if fn == 'mean':
            return a.float().mean() if axis is None else a.float().mean(dim=axis, keepdim=kd)
        if fn == 'max':
            return a.max() if axis is None else a.max(dim=axis, keepdim=kd).values
        raise NotImplementedError(fn)
    if op == 'flatten':
        return _ev(I[0], c).reshape(-1)
    if op == 'reshape':
        a = _ev(I[0], c)
        return a.reshape([int(_ev(s, c)) for s in I[1:]])
    if op == 'eye':
        return torch.eye(int(_ev(I[0], c)), int(_ev(I[1], c)), dtype=torch.float32)
    if op == 'concatenate':
        parts = [_ev(p, c) for p in I]
        if any(p.is_floating_point() for p in parts):
            parts = [p.float() for p in parts]
        try:
            return torch.cat(parts, dim=v.attrs['axis'])
        except RuntimeError as e:
            raise RuntimeError('%s ; parts: %s ; ops: %s' % (e, [tuple(p.shape) for p in parts], [(q.op, q.attrs.get('spec'), [tuple(_ev(z, c).shape) if z is not None else None for z in q.inputs]) for q in I]))
    if op == 'diag':
        return torch.diagonal(_ev(I[0], c))
    if op == 'subtensor':
        return _ev(I[0], c)[_build_index(v, c)]
    if op in ('set_subtensor', 'inc_subtensor'):
        # x = base[idx] ; result = base with x replaced / incremented.  Not differentiated through (the
        # reference only uses it inside `updates`).  Duplicate-index semantics: NumPy order.
        sub = I[0]
        base = _ev(sub.inputs[0], c).detach().numpy().copy()
        index = _build_index(sub, c, numpy=True)
        y = _ev(I[1], c).detach().numpy()
        if op == 'set_subtensor':
            base[index] = y
        else:
            np.add.at(base, index, y)
        return torch.from_numpy(base)
    if op == 'random':
        key = v.serial
        if key not in c.rng_cache:
            shape = tuple(int(_ev(s, c)) for s in I)
            rank = _rng_nodes.index(v.serial)
            if RNG_HOOK[0] is None:
                raise RuntimeError('theano shim: no RNG hook installed')
            c.rng_cache[key] = _to_t(RNG_HOOK[0](v.attrs['kind'], rank, c.call_count, shape, v.attrs))
        return c.rng_cache[key]
    if op == 'searchsorted_gpu':
        # GpuBinarySearchSorted, custom_theano_ops.py:318-349: upper bound with end clamps
        P = _ev(I[0], c).numpy()
        x = _ev(I[1], c).numpy()
        out = np.searchsorted(P, x, side='right').astype(np.int64)
        out[x > P[-1]] = len(P)
        out[x <= P[0]] = 0
        return torch.from_numpy(out)
    if op == 'extract_diag':
        # GpuExtractDiag2D, custom_theano_ops.py:66-78 (main diagonal, optional keepdims)
        d = torch.diagonal(_ev(I[0], c))
        return d[:, None] if v.attrs['keepdims'] else d
    if op == 'grad':
        raise RuntimeError('grad nodes are resolved by Function')
    raise NotImplementedError(op)


def _build_index(v, c, numpy=False):
    out = []
    for kind, ref in v.attrs['spec']:
        if kind == 'slice':
            parts = [None if p is None else int(_ev(v.inputs[p], c)) for p in ref]
            out.append(slice(*parts))
        else:
            t = _idx_of(v.inputs[ref], c)
            if isinstance(t, torch.Tensor) and t.ndim == 0:
                out.append(int(t))
            else:
                out.append(t.numpy() if numpy else t)
    return tuple(out) if len(out) > 1 else out[0]


def _collect(v, seen, out):
    if v is None or v.serial in seen:
        return
    seen.add(v.serial)
    out.append(v)
    for i in v.inputs:
        _collect(i, seen, out)
    if v.op == 'grad':
        _collect(v.attrs['cost'], seen, out)
        _collect(v.attrs['wrt'], seen, out)


class Function:
    def __init__(self, inputs, outputs=None, updates=None, allow_input_downcast=False, on_unused_input=None,
                 **kw):
        self.inputs = list(inputs)
        self.single = not isinstance(outputs, (list, tuple))
        self.outputs = [] if outputs is None else ([outputs] if self.single else list(outputs))
        self.no_out = outputs is None
        self.updates = [(k, _as_var(u)) for k, u in (updates.items() if updates else [])]
        nodes, seen = [], set()
        for o in self.outputs + [u for _, u in self.updates]:
            _collect(o, seen, nodes)
        self.grads = [n for n in nodes if n.op == 'grad']
        self.calls = 0

    def __call__(self, *args):
        env = {}
        for v, a in zip(self.inputs, args):
            env[v.serial] = _to_t(np.asarray(a))
        rng_cache = {}
        c = _Ctx(env, self.calls, rng_cache)
        if self.grads:
            # one differentiation per distinct cost: the wrt nodes become autograd leaves
            by_cost = {}
            for g in self.grads:
                by_cost.setdefault(g.attrs['cost'].serial, []).append(g)
            for _, gs in by_cost.items():
                cg = _Ctx(env, self.calls, rng_cache)
                leaves = []
                for g in gs:
                    w = g.attrs['wrt']
                    if w.serial not in cg.leaf:
                        val = _ev(w, c).detach().clone().float().requires_grad_(True)
                        cg.leaf[w.serial] = val
                    leaves.append(cg.leaf[w.serial])
                cost = _ev(gs[0].attrs['cost'], cg)
                res = torch.autograd.grad(cost, leaves, allow_unused=True)
                for g, r, lf in zip(gs, res, leaves):
                    c.memo[g.serial] = torch.zeros_like(lf) if r is None else r.detach()
        with torch.no_grad():
            outs = [_ev(o, c) for o in self.outputs]
            new = [(k, _ev(u, c)) for k, u in self.updates]
        for k, val in new:
            a = val.detach().numpy() if isinstance(val, torch.Tensor) else np.asarray(val)
            k.value = np.array(a, dtype=k.value.dtype).reshape(a.shape)
        self.calls += 1
        if self.no_out:
            CALL_LOG.append((id(self), None))
            return None
        res = [o.detach().numpy() if isinstance(o, torch.Tensor) else np.asarray(o) for o in outs]
        CALL_LOG.append((id(self), res[0] if self.single else res))
        return res[0] if self.single else res


def function(inputs=(), outputs=None, updates=None, **kw):
    return Function(inputs, outputs, updates, **kw)


from . import tensor  # noqa: E402,F401
