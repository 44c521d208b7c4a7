"""Host-side helpers mirroring gpitch/methods.py (nonlinearities :197-233, midi2freq :266-270)."""
import numpy as np

from . import _lib


def norm(x):
    """methods.py:193-195"""
    return x / np.max(np.abs(x))


def logistic(x):
    """methods.py:197-199 — NOTE shifted and scaled: 1/(1+exp(-2(x-pi)))"""
    return 1. / (1. + np.exp(-2. * (x - np.pi)))


def ilogistic(x):
    """methods.py:201-203"""
    return - np.log(1. / x - 1.)


def softplus(x):
    """methods.py:205-207"""
    return np.log(np.exp(x) + 1.)


def isoftplus(x):
    """methods.py:209-211"""
    return np.log(np.exp(x) - 1.)


def gaussfun(x):
    """methods.py:213-214"""
    return np.exp(-2. * (x - np.pi) ** 2)


class _Nonlinearity(object):
    """Stand-in for the reference's `*_tf` graph functions (methods.py:216-233): callable on host
    arrays, and carries the code the HIP likelihood kernel switches on."""

    def __init__(self, name, code, fn):
        self.__name__ = name
        self.code = code
        self._fn = fn

    def __call__(self, x):
        return self._fn(np.asarray(x, dtype=np.float64))

    def __repr__(self):
        return "<gpitch_amd nonlinearity %s>" % self.__name__


logistic_tf = _Nonlinearity("logistic_tf", _lib.NLIN_LOGISTIC, logistic)
softplus_tf = _Nonlinearity("softplus_tf", _lib.NLIN_SOFTPLUS, softplus)
gaussfun_tf = _Nonlinearity("gaussfun_tf", _lib.NLIN_GAUSS, gaussfun)


def nlin_code(nlinfun):
    code = getattr(nlinfun, "code", None)
    if code is None:
        raise TypeError("nlinfun must be one of gpitch_amd.logistic_tf / softplus_tf / gaussfun_tf "
                        "(arbitrary Python callables cannot run inside the HIP likelihood kernel)")
    return code


def midi2freq(midi):
    """methods.py:266-267"""
    return 2. ** ((midi - 69.) / 12.) * 440.


def freq2midi(freq):
    """methods.py:269-270"""
    return int(69. + 12. * np.log2(freq / 440.))
