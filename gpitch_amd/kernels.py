"""Kernel operator API: Kern.K(X, X2=None, presliced=False) / Kern.Kdiag(X), as the reference's
kernels expose it to GPflow (gpitch/matern12_spectral_mixture.py:38,58,102,119) plus the GPflow-0.5
stationary kernels the reference instantiates (Matern32 init_kernels.py:12, demo-modgp.py:32;
Matern12 init_models.py:83; Matern52 init_models.py:188; RBF) and the `Add` kernel built by
`k1 + k2` / np.sum(list) (transcription.py:245, sgpr_ss.py:19-22).  The arithmetic runs in
csrc/cov.hip through gp_kernel_build / gp_kernel_diag."""
import ctypes as C

import numpy as np

from . import _lib
from .param import Param, ParamList, Parameterized, transforms


class Kern(Parameterized):
    type_code = None

    def __init__(self, input_dim, active_dims=None):
        if input_dim != 1:
            raise ValueError("gpitch kernels act on 1-D time inputs (input_dim=1)")
        self.input_dim = input_dim
        self.active_dims = active_dims

    # --- device descriptor -----------------------------------------------------------------
    num_partials = 0   # spectral-mixture kernels overwrite this per instance

    def theta(self):
        """constrained hyper-parameter vector in the C-ABI layout [variance, lengthscales, e.., f..]"""
        raise NotImplementedError

    def theta_params(self):
        """the Param objects behind theta(), same order"""
        raise NotImplementedError

    def oracle_dict(self):
        """plain-dict view used by tests to feed the CPU oracle the same hyper-parameters"""
        raise NotImplementedError

    def _slice(self, X, X2):
        X = np.asarray(X, dtype=np.float64).reshape(-1, 1)
        X2 = None if X2 is None else np.asarray(X2, dtype=np.float64).reshape(-1, 1)
        return X, X2

    def K(self, X, X2=None, presliced=False):
        h = _lib.default_handle()
        X, X2 = self._slice(X, X2)
        n1 = X.shape[0]
        n2 = n1 if X2 is None else X2.shape[0]
        th = h.to_device(self.theta())
        d = _lib.KernelDesc(self.type_code, self.num_partials, th.data_ptr())
        dx1 = h.to_device(X)
        dx2 = None if X2 is None else h.to_device(X2)
        out = h.empty(n1, n2)
        h.check(h.lib.gp_kernel_build(h.h, C.byref(d), dx1.data_ptr(), n1, None if dx2 is None else dx2.data_ptr(),
                                      n2, out.data_ptr(), n2, 0))
        return out.cpu().numpy()

    def Kdiag(self, X, presliced=False):
        h = _lib.default_handle()
        n = np.asarray(X).reshape(-1, 1).shape[0]
        th = h.to_device(self.theta())
        d = _lib.KernelDesc(self.type_code, self.num_partials, th.data_ptr())
        out = h.empty(n)
        h.check(h.lib.gp_kernel_diag(h.h, C.byref(d), n, out.data_ptr(), 0))
        return out.cpu().numpy()

    def __add__(self, other):
        return Add([self, other])

    def __radd__(self, other):
        if other == 0:  # np.sum(list_of_kernels) starts from 0
            return self
        return Add([other, self])


class Stationary(Kern):
    """GPflow 0.5 Stationary: variance / lengthscales (positive), r = sqrt(|x-x'|^2/l^2 + 1e-12)."""
    oracle_name = None

    def __init__(self, input_dim, variance=1.0, lengthscales=None, active_dims=None, ARD=False):
        Kern.__init__(self, input_dim, active_dims)
        if ARD:
            raise ValueError("ARD is not used on the gpitch path")
        self.ARD = False
        self.variance = Param(variance, transforms.positive)
        self.lengthscales = Param(1.0 if lengthscales is None else lengthscales, transforms.positive)

    def theta(self):
        return np.array([self.variance.value[0], self.lengthscales.value[0]])

    def theta_params(self):
        return [self.variance, self.lengthscales]

    def oracle_dict(self):
        return {"type": self.oracle_name, "variance": float(self.variance.value[0]),
                "lengthscales": float(self.lengthscales.value[0]), "energy": [], "frequency": []}


class Matern12(Stationary):
    type_code = _lib.KERN_MATERN12
    oracle_name = "matern12"


class Matern32(Stationary):
    type_code = _lib.KERN_MATERN32
    oracle_name = "matern32"


class Matern52(Stationary):
    type_code = _lib.KERN_MATERN52
    oracle_name = "matern52"


class RBF(Stationary):
    type_code = _lib.KERN_RBF
    oracle_name = "rbf"


class Add(Kern):
    """GPflow Add kernel: K = sum_p K_p; exposes .kern_list (sgpr_ss.py:19-22,86)."""

    def __init__(self, kern_list):
        Kern.__init__(self, 1)
        flat = []
        for k in kern_list:
            flat.extend(k.kern_list if isinstance(k, Add) else [k])
        self.kern_list = flat
        self._kern_params = ParamList(flat)

    def K(self, X, X2=None, presliced=False):
        out = self.kern_list[0].K(X, X2)
        for k in self.kern_list[1:]:
            out = out + k.K(X, X2)
        return out

    def Kdiag(self, X, presliced=False):
        out = self.kern_list[0].Kdiag(X)
        for k in self.kern_list[1:]:
            out = out + k.Kdiag(X)
        return out
