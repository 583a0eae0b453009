"""`theano.tensor` subset used by /root/reference/gru4rec.py, gpu_ops.py and evaluation.py (see theano/__init__.py)."""
import numpy as np

from .. import Var, _as_var, config
from . import nnet  # noqa: F401


def _inp(ndim, dtype, name=None):
    return Var('input', [], ndim=ndim, name=name, dtype=dtype)


def ivector(name=None): return _inp(1, 'int32', name)
def iscalar(name=None): return _inp(0, 'int32', name)
def bcol(name=None): return _inp(2, 'int8', name)
def fmatrix(name=None): return _inp(2, 'float32', name)


def dot(a, b):
    a, b = _as_var(a), _as_var(b)
    return Var('dot', [a, b], ndim=(a.ndim or 0) + (b.ndim or 0) - 2)


def _e1(fn, x, **kw):
    x = _as_var(x)
    return Var('elem1', [x], fn=fn, ndim=x.ndim, **kw)


def exp(x): return _e1('exp', x)
def log(x): return _e1('log', x)
def sqrt(x): return _e1('sqrt', x)
def tanh(x): return _e1('tanh', x)
def cast(x, dtype): return _e1('cast', x, to=str(dtype), dtype=str(dtype))
def zeros_like(x, dtype=None): return _e1('zeros_like', x, to=dtype)
def ones_like(x, dtype=None): return _e1('ones_like', x, to=dtype)


def maximum(a, b):
    a, b = _as_var(a), _as_var(b)
    return Var('elem2', [a, b], fn='maximum', ndim=max(a.ndim or 0, b.ndim or 0))


def ge(a, b): return _as_var(a) >= b
def gt(a, b): return _as_var(a) > b


def switch(c, a, b):
    c, a, b = _as_var(c), _as_var(a), _as_var(b)
    return Var('switch', [c, a, b], ndim=max(c.ndim or 0, a.ndim or 0, b.ndim or 0))


def eye(n, m=None):
    return Var('eye', [_as_var(n), _as_var(n if m is None else m)], ndim=2)


def concatenate(parts, axis=0):
    parts = [_as_var(p) for p in parts]
    return Var('concatenate', parts, axis=axis, ndim=parts[0].ndim)


def sum(x, axis=None, keepdims=False):  # noqa: A001
    if isinstance(x, (list, tuple)) and any(isinstance(e, Var) for e in x):
        # T.sum of a Python list of scalars (gru4rec.py:387): the list is first stacked by as_tensor_variable
        total = None
        for e in x:
            e = _as_var(e)
            total = e if total is None else total + e
        return total
    return _as_var(x).sum(axis=axis, keepdims=keepdims)


def mean(x, axis=None, keepdims=False):
    return _as_var(x).mean(axis=axis, keepdims=keepdims)


def diag(x):
    return Var('diag', [_as_var(x)], ndim=1)


def grad(cost, wrt):
    """d cost / d wrt ; wrt may be a shared variable or any intermediate expression of the graph."""
    if isinstance(wrt, (list, tuple)):
        return [grad(cost, w) for w in wrt]
    w = _as_var(wrt)
    return Var('grad', [], cost=cost, wrt=w, ndim=w.ndim)


def set_subtensor(x, y):
    assert x.op == 'subtensor'
    return Var('set_subtensor', [x, _as_var(y)], ndim=x.inputs[0].ndim)


def inc_subtensor(x, y):
    assert x.op == 'subtensor'
    return Var('inc_subtensor', [x, _as_var(y)], ndim=x.inputs[0].ndim)
