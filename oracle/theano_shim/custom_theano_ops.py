"""Stand-in for the reference's custom_theano_ops.py (which needs theano.gpuarray / pygpu): the two ops that
gpu_ops.py dispatches to, with the semantics of their CUDA kernels (custom_theano_ops.py:66-78, :318-349)."""
from theano import Var, _as_var


class GpuExtractDiag2D:
    def __init__(self, offset=0, keepdims=False, **kw):
        self.keepdims = keepdims

    def __call__(self, x):
        x = _as_var(x)
        return Var('extract_diag', [x], keepdims=self.keepdims, ndim=2 if self.keepdims else 1)


class GpuBinarySearchSorted:
    def __init__(self, dtype_int64=True, **kw):
        pass

    def __call__(self, d, x):
        return Var('searchsorted_gpu', [_as_var(d), _as_var(x)], ndim=1, dtype='int64')
