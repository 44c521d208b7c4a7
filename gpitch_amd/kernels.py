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

    def K(self, X, X2=None, presliced=False, float_type=None):
        """float_type=np.float32: a float32 result (gp_kernel_build_f32; rows padded to a multiple of 4 floats on the
        device, as the engine's Kuf strips are)"""
        h = _lib.default_handle()
        X, X2 = self._slice(X, X2)
        n1 = X.shape[0]
        n2 = n1 if X2 is None else X2.shape[0]
        th = h.to_device(self.theta())
        d = _lib.KernelDesc(self.type_code, self.num_partials, th.data_ptr())
        dx1 = h.to_device(X)
        dx2 = None if X2 is None else h.to_device(X2)
        if _lib.precision_bits(float_type) == 32:
            ld = (n2 + 3) // 4 * 4
            out32 = h.torch.zeros(n1, ld, dtype=h.torch.float32, device=h.device)
            h.check(h.lib.gp_kernel_build_f32(h.h, C.byref(d), dx1.data_ptr(), n1,
                                              None if dx2 is None else dx2.data_ptr(), n2, out32.data_ptr(), ld, 0))
            return out32[:, :n2].cpu().numpy()
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

    def __mul__(self, other):
        return Prod(self, other)

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


class Matern32sm(Kern):
    """Matern spectral mixture kernel with single lengthscale (gpitch/kernels.py:204-258; init_models.py:84,96):
    K = (1 + r1) exp(-r1) sum_k variance_k cos(2 pi f_k r),  r = |x - x' + 1e-12|,  r1 = sqrt(3) r / l.
    lengthscales ~ Logistic(0, 2), variance_k ~ Logistic(0, 0.25), frequency_k positive (:219-223)."""
    type_code = _lib.KERN_MATERN32SM
    oracle_name = "matern32sm"

    def __init__(self, input_dim, num_partials, lengthscales=None, variances=None, frequencies=None):
        Kern.__init__(self, input_dim, active_dims=None)
        self.ARD = False
        self.num_partials = int(num_partials)
        if lengthscales is None:      # kernels.py:215-218
            lengthscales = 1.
            variances = 0.125 * np.ones((num_partials, 1))
            frequencies = 1. * (1. + np.arange(num_partials))
        variances = np.asarray(variances, dtype=np.float64).reshape(-1)
        frequencies = np.asarray(frequencies, dtype=np.float64).reshape(-1)
        self.lengthscales = Param(lengthscales, transforms.Logistic(0., 2.))
        self.variance = ParamList([Param(variances[i], transforms.Logistic(0., 0.25)) for i in range(self.num_partials)])
        self.frequency = ParamList([Param(frequencies[i], transforms.positive) for i in range(self.num_partials)])
        self._unit = Param(1.0)       # the C-ABI layout keeps a global variance slot: 1, never trained
        self._unit.fixed = True

    def vars_n_freqs_fixed(self, fix_var=True, fix_freq=False):
        """kernels.py:255-258"""
        for i in range(self.num_partials):
            self.variance[i].fixed = fix_var
            self.frequency[i].fixed = fix_freq

    def theta(self):
        return np.concatenate([[1.0, self.lengthscales.value[0]], [v.value[0] for v in self.variance],
                               [f.value[0] for f in self.frequency]])

    def theta_params(self):
        return [self._unit, self.lengthscales] + list(self.variance) + list(self.frequency)

    def oracle_dict(self):
        return {"type": self.oracle_name, "variance": 1.0, "lengthscales": float(self.lengthscales.value[0]),
                "energy": [float(v.value[0]) for v in self.variance],
                "frequency": [float(f.value[0]) for f in self.frequency]}


class MercerCosMix(Kern):
    """The Mercer Cosine Mixture kernel (gpitch/kernels.py:321-376): K = variance * Phi(x)^T Phi(x'),
    Kdiag = variance.  On the gpitch path it only appears as a factor of Matern52 * MercerCosMix
    (init_models.py:183-198), which is what the engine implements (see Prod)."""

    def __init__(self, input_dim, energy=np.asarray([1.]), frequency=np.asarray([2 * np.pi]), variance=1.0,
                 features_as_params=False):
        Kern.__init__(self, input_dim, active_dims=None)
        self.num_features = len(frequency)
        self.variance = Param(variance, transforms.Logistic(0., 0.25))
        energy = np.asarray(energy, dtype=np.float64).reshape(-1)
        frequency = np.asarray(frequency, dtype=np.float64).reshape(-1)
        self.energy = ParamList([Param(e, transforms.positive) for e in energy])
        self.frequency = ParamList([Param(f, transforms.positive) for f in frequency])
        if not features_as_params:      # plain arrays in the reference: never trained
            self.energy.fixed = True
            self.frequency.fixed = True


class Prod(Kern):
    """k1 * k2 for the one product the reference builds: Matern52 * MercerCosMix (init_models.py:188-193)
    = v52 v_c (1 + sqrt5 r + 5/3 r^2) exp(-sqrt5 r) sum_k e_k cos(2 pi f_k (x - x')).
    The engine carries the product v52 * v_c as one (fixed) variance: the reference fixes both factors' variances
    (init_models.py:189,192); the Matern52 lengthscale and, with features_as_params, energies / frequencies train."""
    type_code = _lib.KERN_MERCER_MATERN52SM
    oracle_name = "mercer_matern52sm"

    def __init__(self, k1, k2):
        Kern.__init__(self, 1)
        if isinstance(k2, Matern52) and isinstance(k1, MercerCosMix):
            k1, k2 = k2, k1
        if not (isinstance(k1, Matern52) and isinstance(k2, MercerCosMix)):
            raise NotImplementedError("only Matern52 * MercerCosMix products are used on the gpitch path")
        self.kern_list = [k1, k2]
        self._kern_params = ParamList([k1, k2])
        self.num_partials = k2.num_features
        self._variance = Param(1.0, transforms.positive)

    def _sync_variance(self):
        k1, k2 = self.kern_list
        if not (k1.variance.fixed and k2.variance.fixed):
            # a trainable factor variance would need its own transform slot; the reference fixes both
            raise NotImplementedError("Matern52 * MercerCosMix: both variances must be fixed (init_models.py:189,192)")
        self._variance.value = k1.variance.value * k2.variance.value
        self._variance.fixed = True
        return self._variance

    def theta(self):
        k1, k2 = self.kern_list
        return np.concatenate([[self._sync_variance().value[0], k1.lengthscales.value[0]],
                               [e.value[0] for e in k2.energy], [f.value[0] for f in k2.frequency]])

    def theta_params(self):
        k1, k2 = self.kern_list
        return [self._sync_variance(), k1.lengthscales] + list(k2.energy) + list(k2.frequency)

    def oracle_dict(self):
        k1, k2 = self.kern_list
        return {"type": self.oracle_name, "variance": float(self._sync_variance().value[0]),
                "lengthscales": float(k1.lengthscales.value[0]),
                "energy": [float(e.value[0]) for e in k2.energy], "frequency": [float(f.value[0]) for f in k2.frequency]}


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
