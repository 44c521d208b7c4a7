"""SGPRSS — sparse GP regression for source separation (gpitch/sgpr_ss.py:10-114)."""
import numpy as np

from .param import DataHolder, Param, ParamList, Parameterized, transforms


class SGPRSS(Parameterized):
    """Same constructor as the reference (sgpr_ss.py:14): SGPRSS(X, Y, kern, Z, mean_function=None, reg=False),
    `kern` a GPflow-style Add kernel with `.kern_list`."""

    def __init__(self, X, Y, kern, Z, mean_function=None, reg=False):
        if mean_function is not None:
            raise NotImplementedError("only the zero mean function is used on the gpitch path")
        if reg:
            kern.var_vector = ParamList([k.variance for k in kern.kern_list])   # sgpr_ss.py:17-22
        self.X = DataHolder(np.asarray(X, dtype=np.float64).reshape(-1, 1))
        self.Y = DataHolder(np.asarray(Y, dtype=np.float64).reshape(-1, 1))
        self.Z = DataHolder(np.asarray(Z, dtype=np.float64).reshape(-1, 1), on_shape_change='pass')
        self.kern = kern
        self.reg = reg
        self.likelihood = _Gaussian()
        self.num_latent = 1

    def build_likelihood(self):
        raise NotImplementedError("SGPRSS bound: HIP path lands next (gp_sgpr_bound)")


class _Gaussian(Parameterized):
    def __init__(self):
        self.variance = Param(1.0, transforms.positive)
