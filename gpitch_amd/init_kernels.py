"""Kernel initialisers — gpitch/init_kernels.py:6-47."""
from .kernels import Matern32
from .matern12_spectral_mixture import MercerMatern12sm


def init_kern_act(num_pitches):
    """Initialize kernels for activations (init_kernels.py:6-13): Matern32(l=1.0, v=3.5)."""
    return [Matern32(1, lengthscales=1.0, variance=3.5) for _ in range(num_pitches)]


def init_kern_com(num_pitches, lengthscale, energy, frequency, len_fixed=True):
    """Initialize component kernels (init_kernels.py:16-38)."""
    kern_com = []
    for i in range(num_pitches):
        kern_com.append(MercerMatern12sm(1, variance=1., lengthscales=lengthscale[i].copy(), energy=energy[i].copy(),
                                         frequency=frequency[i].copy(), len_fixed=len_fixed))
    return kern_com


def init_kern(num_pitches, lengthscale, energy, frequency):
    """init_kernels.py:41-47"""
    return [init_kern_act(num_pitches), init_kern_com(num_pitches, lengthscale, energy, frequency)]
