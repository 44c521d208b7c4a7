"""Array backends for the oracle (TEST INFRASTRUCTURE ONLY — see oracle/__init__.py).

The restatement in ``gpflow05.py`` is written once against this tiny interface so
that the numpy evaluation (golden values) and the torch-CPU evaluation (autograd
gradients, CPU baseline timing) cannot drift apart.
"""
import numpy as np


class NumpyBackend(object):
    name = "numpy"
    pi = np.pi

    def asarray(self, a):
        return np.asarray(a, dtype=np.float64)

    def scalar(self, a):
        return np.float64(a)

    # elementwise
    sqrt = staticmethod(np.sqrt)
    exp = staticmethod(np.exp)
    log = staticmethod(np.log)
    cos = staticmethod(np.cos)
    sin = staticmethod(np.sin)
    square = staticmethod(np.square)
    abs = staticmethod(np.abs)

    def sum(self, a, axis=None):
        return np.sum(a, axis=axis)

    def matmul(self, a, b):
        return np.matmul(a, b)

    def t(self, a):
        return a.T

    def eye(self, n):
        return np.eye(n)

    def tril(self, a):
        return np.tril(a)

    def diag_part(self, a):
        return np.diagonal(a)

    def cholesky(self, a):
        return np.linalg.cholesky(a)

    def trsm(self, L, B, lower=True):
        from scipy.linalg import solve_triangular
        return solve_triangular(L, B, lower=lower, check_finite=False)

    def stack(self, lst, axis=0):
        return np.stack(lst, axis=axis)

    def concat(self, lst, axis=0):
        return np.concatenate(lst, axis=axis)

    def reshape(self, a, shape):
        return np.reshape(a, shape)

    def fill(self, n, value):
        return np.full((n,), value, dtype=np.float64)

    def zeros_like(self, a):
        return np.zeros_like(a)


class TorchBackend(object):
    """torch-CPU float64, differentiable (mirrors TF reverse-mode for the baseline)."""
    name = "torch"
    pi = np.pi

    def __init__(self):
        import torch
        self.torch = torch

    def asarray(self, a):
        torch = self.torch
        if isinstance(a, torch.Tensor):
            return a.to(torch.float64)
        return torch.as_tensor(np.asarray(a, dtype=np.float64))

    def scalar(self, a):
        return self.asarray(a)

    def sqrt(self, a): return self.torch.sqrt(a)
    def exp(self, a): return self.torch.exp(a)
    def log(self, a): return self.torch.log(a)
    def cos(self, a): return self.torch.cos(a)
    def sin(self, a): return self.torch.sin(a)
    def square(self, a): return a * a
    def abs(self, a): return self.torch.abs(a)

    def sum(self, a, axis=None):
        return a.sum() if axis is None else a.sum(dim=axis)

    def matmul(self, a, b):
        return a @ b

    def t(self, a):
        return a.transpose(-1, -2)

    def eye(self, n):
        return self.torch.eye(n, dtype=self.torch.float64)

    def tril(self, a):
        return self.torch.tril(a)

    def diag_part(self, a):
        return self.torch.diagonal(a)

    def cholesky(self, a):
        return self.torch.linalg.cholesky(a)

    def trsm(self, L, B, lower=True):
        return self.torch.linalg.solve_triangular(L, B, upper=not lower)

    def stack(self, lst, axis=0):
        return self.torch.stack(lst, dim=axis)

    def concat(self, lst, axis=0):
        return self.torch.cat(lst, dim=axis)

    def reshape(self, a, shape):
        return a.reshape(shape)

    def fill(self, n, value):
        return self.torch.ones(n, dtype=self.torch.float64) * value

    def zeros_like(self, a):
        return self.torch.zeros_like(a)


NP = NumpyBackend()
