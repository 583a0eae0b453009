"""MRG_RandomStreams stand-in: every draw is delegated to theano.RNG_HOOK (see theano/__init__.py)."""
from .. import Var, _as_var, _rng_nodes


class MRG_RandomStreams:
    def __init__(self, seed=12345, **kw):
        self.seed = seed

    def _node(self, kind, size, **attrs):
        shape = list(size) if not isinstance(size, Var) else [size]
        v = Var('random', [_as_var(s) for s in shape], kind=kind, ndim=len(shape), **attrs)
        _rng_nodes.append(v.serial)
        return v

    def uniform(self, size, low=0.0, high=1.0, dtype='float32', **kw):
        return self._node('uniform', size)

    def binomial(self, size, n=1, p=0.5, dtype='float32', **kw):
        return self._node('binomial', size, p=p)
