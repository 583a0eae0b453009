from .. import Var, _as_var


def sigmoid(x):
    x = _as_var(x)
    return Var('elem1', [x], fn='sigmoid', ndim=x.ndim)
