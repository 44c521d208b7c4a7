"""GPflow-0.5 mean functions as they can be handed to SGPRSS(mean_function=...) (gpitch/sgpr_ss.py:14,25,40,90-99).

The reference passes `mean_function` through to gpflow.sgpr.SGPR, subtracts it from the data in the bound
(`err = Y - mean_function(X)`, :40) and in the exact per-source posterior (:90), and adds it back to the predicted
means (:95).  None of its callers sets one (transcription.py:245, separation.py:257 use the default Zero).
GPflow's Constant / Linear carry trainable Params (`c`; `A`, `b`): they are Params here too, SGPRSS.optimize() trains them
with the kernel hyper-parameters unless `.fixed = True` — the bound's derivative with respect to err comes from the device
(include/gpitch_abi.h: gp_sgpr_residual_grad), the chain rule through the mean function is `grad_from_residual`.
Anything else callable on an (n, 1) array that returns (n, 1) works as a fixed function of the inputs."""
import numpy as np

from .param import Param


class Zero(object):
    def __call__(self, X):
        return np.zeros((np.asarray(X).reshape(-1, 1).shape[0], 1))


class Constant(object):
    """gpflow.mean_functions.Constant: m(x) = c"""
    def __init__(self, c=0.0):
        self.c = Param(float(np.asarray(c).reshape(-1)[0]))

    def params(self):
        return [self.c]

    def __call__(self, X):
        return np.full((np.asarray(X).reshape(-1, 1).shape[0], 1), self.c.value[0])

    def grad_from_residual(self, X, r):
        """d bound / d c given r = d bound / d err (err = Y - m(X))"""
        return [np.array([-np.sum(r)])]


class Linear(object):
    """gpflow.mean_functions.Linear for one input and one output column: m(x) = A x + b"""
    def __init__(self, A=1.0, b=0.0):
        self.A = Param(float(np.asarray(A).reshape(-1)[0]))
        self.b = Param(float(np.asarray(b).reshape(-1)[0]))

    def params(self):
        return [self.A, self.b]

    def __call__(self, X):
        return np.asarray(X, dtype=np.float64).reshape(-1, 1) * self.A.value[0] + self.b.value[0]

    def grad_from_residual(self, X, r):
        x = np.asarray(X, dtype=np.float64).reshape(-1)
        r = np.asarray(r, dtype=np.float64).reshape(-1)
        return [np.array([-np.dot(r, x)]), np.array([-np.sum(r)])]
