"""Inducing-point initialisers — gpitch/init_models.py:9-71 (host numpy, O(N))."""
import numpy as np
from scipy import signal


def _hann(n):
    # scipy.signal.hann of the reference's era == scipy.signal.windows.hann (symmetric)
    return signal.windows.hann(n)


def init_liv(x, y, num_sources=1, win_size=9, thres=0.0025, dec=1):
    """Initialize location of inducing variables from the extrema of the data (init_models.py:9-51).

    Faithful to the reference including its quirk at :38-43: `np.argsort` is applied to the *tuple*
    returned by np.where, which yields [[0..k-1]] — so the first k extrema are returned (k = number
    above the energy threshold), not the k above threshold."""
    x = x.reshape(-1, )
    y = y.reshape(-1, )
    win1 = _hann(1600)
    energy = signal.convolve(np.abs(y), win1, mode='same') / sum(win1)
    energy /= np.max(energy)
    win2 = _hann(win_size)
    y_smooth = signal.convolve(y, win2, mode='same') / sum(win2)
    f_sign = np.sign(np.gradient(y_smooth))
    f_change_sign = np.diff(f_sign)
    idx = np.where(f_change_sign)
    x_all = x[idx].copy()
    y_all = y[idx].copy()
    energy_all = energy[idx].copy()
    idx1 = np.where(energy_all > thres)
    idx3 = np.argsort(idx1)
    x_final = x_all[idx3].copy().reshape(-1, 1)
    y_final = y_all[idx3].copy().reshape(-1, 1)
    za, zc = [], []
    for i in range(num_sources):
        za.append(x_final[::dec].copy())
        zc.append(x_final[::dec].copy())
    return [za, zc], y_final[::dec]


def init_iv(x, num_sources, nivps_a, nivps_c, fs):
    """Uniform inducing variables (init_models.py:54-71); Python-2 integer division made explicit."""
    za, zc = [], []
    dec_a = int(fs // nivps_a)
    dec_c = int(fs // nivps_c)
    for i in range(num_sources):
        za.append(np.vstack([x[::dec_a].copy(), x[-1].copy()]))
        zc.append(np.vstack([x[::dec_c].copy(), x[-1].copy()]))
    return [za, zc]
