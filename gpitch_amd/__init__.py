"""gpitch_amd — MI355X-native engine behind the gpitch pdgp / sgpr_ss ELBO path.

Importing the package loads libgpitch_hip.so (the hand-written HIP kernels behind a C-ABI,
include/gpitch_abi.h).  There is no CPU fallback: a missing library raises ImportError here and a
missing gfx950 device raises when the first handle is created.
"""
from . import _lib

_lib.load_library()

from .methods import (logistic, ilogistic, softplus, isoftplus, gaussfun, logistic_tf, softplus_tf,  # noqa: E402
                      gaussfun_tf, midi2freq, freq2midi, norm, find_ideal_f0, readaudio, init_cparam)
from . import param, kernels, matern12_spectral_mixture, likelihoods, conditionals, pdgp, sgpr_ss, synth, train  # noqa: E402,F401
from .init_models import init_liv, init_iv  # noqa: E402,F401
from .window_overlap import segmented, windowed, merged_mean, merged_variance  # noqa: E402,F401
from .init_kernels import init_kern_act, init_kern_com, init_kern  # noqa: E402,F401
