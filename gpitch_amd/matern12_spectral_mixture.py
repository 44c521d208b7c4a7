"""Matérn-1/2 spectral-mixture kernels — same class names, constructor arguments and parameter
semantics as gpitch/matern12_spectral_mixture.py (Matern12sm :14-67, MercerMatern12sm :70-133).
K / Kdiag run in csrc/cov.hip (feature form Phi^T Phi for the Mercer kernel, broadcast cosine form
for Matern12sm)."""
import numpy as np

from . import _lib
from .kernels import Kern, Stationary
from .param import Param, ParamList, transforms


class _SpectralMixtureMixin(object):
    @property
    def num_partials_(self):
        return len(self.frequency)

    def theta(self):
        return np.concatenate([[self.variance.value[0], self.lengthscales.value[0]],
                               [e.value[0] for e in self.energy], [f.value[0] for f in self.frequency]])

    def theta_params(self):
        return [self.variance, self.lengthscales] + list(self.energy) + list(self.frequency)

    def oracle_dict(self):
        return {"type": self.oracle_name, "variance": float(self.variance.value[0]),
                "lengthscales": float(self.lengthscales.value[0]),
                "energy": [float(e.value[0]) for e in self.energy],
                "frequency": [float(f.value[0]) for f in self.frequency]}


class Matern12sm(_SpectralMixtureMixin, Kern):
    """Matern spectral mixture kernel with single lengthscale (matern12_spectral_mixture.py:14-67);
    energies and frequencies fixed by default (:34)."""
    type_code = _lib.KERN_MATERN12SM
    oracle_name = "matern12sm"

    def __init__(self, input_dim, variance=1., lengthscales=None, energy=None, frequency=None, len_fixed=False):
        Kern.__init__(self, input_dim, active_dims=None)
        self.ARD = False
        self.num_partials = len(energy)
        self.energy = ParamList([Param(energy[i], transforms.positive) for i in range(self.num_partials)])
        self.frequency = ParamList([Param(frequency[i], transforms.positive) for i in range(self.num_partials)])
        self.variance = Param(variance, transforms.positive)
        self.lengthscales = Param(1.0 if lengthscales is None else lengthscales, transforms.positive)
        self.vars_n_freqs_fixed(fix_energy=True, fix_freq=True)
        if len_fixed:
            self.lengthscales.fixed = True

    def vars_n_freqs_fixed(self, fix_energy=True, fix_freq=True):
        """matern12_spectral_mixture.py:64-67"""
        for i in range(self.num_partials):
            self.energy[i].fixed = fix_energy
            self.frequency[i].fixed = fix_freq


class MercerMatern12sm(_SpectralMixtureMixin, Stationary):
    """The Mercer Matern 1/2 spectral mixture kernel (matern12_spectral_mixture.py:70-133); every
    hyper-parameter trainable by default (:86-94)."""
    type_code = _lib.KERN_MERCER_MATERN12SM
    oracle_name = "mercer_matern12sm"

    def __init__(self, input_dim, energy=np.asarray([1.]), frequency=np.asarray([2 * np.pi]), variance=1.,
                 lengthscales=1., len_fixed=False):
        Stationary.__init__(self, input_dim, variance=variance, lengthscales=lengthscales, active_dims=None, ARD=False)
        self.num_partials = len(frequency)
        self.energy = ParamList([Param(energy[i], transforms.positive) for i in range(self.num_partials)])
        self.frequency = ParamList([Param(frequency[i], transforms.positive) for i in range(self.num_partials)])
        if len_fixed:
            self.lengthscales.fixed = True
