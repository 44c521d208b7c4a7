"""Modulated-GP likelihood — gpitch/likelihoods.py:279-447 (MpdLik), with hermgauss1d (:33-45) and
log_lik_exp (:47-68) executed by csrc/lik.hip through gp_mpd_varexp."""
import ctypes as C

import numpy as np

from . import _lib
from .methods import nlin_code
from .param import Param, Parameterized, transforms


class MpdLik(Parameterized):
    '''Modulated GP likelihood'''

    def __init__(self, nlinfun, num_sources):
        self.variance = Param(1., transforms.positive)   # likelihoods.py:283
        self.nlinfun = nlinfun
        self.num_sources = num_sources
        self.num_gauss_hermite_points = 20

    def logp(self, F, Y):
        """likelihoods.py:287-323: log N(y | sum_i nlin(g_i) f_i, variance).  (Not on the ELBO path;
        a host-side formula.)"""
        F = np.asarray(F, dtype=np.float64)
        P = self.num_sources
        mean = np.zeros(F.shape[0])
        for i in range(P):
            mean = mean + self.nlinfun(F[:, i]) * F[:, i + P]
        y = np.asarray(Y, dtype=np.float64)[:, 0]
        v = self.variance.value[0]
        return (-0.5 * np.log(2 * np.pi) - 0.5 * np.log(v) - 0.5 * (y - mean) ** 2 / v).reshape(-1, 1)

    def variational_expectations(self, Fmu, Fvar, Y):
        """likelihoods.py:325-447.  Fmu, Fvar: N x 2P with columns [g_0..g_{P-1}, f_0..f_{P-1}]."""
        h = _lib.default_handle()
        Fmu = np.ascontiguousarray(Fmu, dtype=np.float64)
        Fvar = np.ascontiguousarray(Fvar, dtype=np.float64)
        N = Fmu.shape[0]
        P = self.num_sources
        if Fmu.shape != (N, 2 * P) or Fvar.shape != (N, 2 * P):
            raise ValueError("Fmu/Fvar must be N x 2*num_sources")
        dmu, dvar = h.to_device(Fmu), h.to_device(Fvar)
        dy = h.to_device(np.asarray(Y, dtype=np.float64).reshape(-1))
        nv = h.to_device(self.variance.value)
        out = h.empty(N)
        h.check(h.lib.gp_mpd_varexp(h.h, dmu.data_ptr(), dvar.data_ptr(), dy.data_ptr(), N, P, nlin_code(self.nlinfun),
                                    nv.data_ptr(), out.data_ptr(), None))
        return out.cpu().numpy().reshape(-1, 1)
