"""Inducing-point and kernel initialisers: the host-side producers of the (z, kern) every caller hands to the hot path.

Own numpy implementations of the behaviour of gpitch/init_models.py (init_liv :9-51, init_iv :54-71,
init_kernel_training :74-89, init_kernel_with_trained_models :92-121, get_features :154-180, init_kern :183-198),
quirks included; pinned by tests/test_host_init.py against the oracle's restatement and against the one anchor the
reference prints (109 inducing points on its demo recording, demos/notebooks/demo_modgp-real-audio.ipynb:88,116).
"""
import numpy as np
from scipy import signal


def _hann_average(sig, width):
    """`sig` smoothed by a unit-gain symmetric Hann window of `width` taps (scipy's 'same' convolution, as the
    reference calls it at init_models.py:19-20,24-25)."""
    taps = signal.windows.hann(width)
    return signal.convolve(sig, taps, mode="same") / taps.sum()


def _turning_points(sig):
    """indices i where the slope of `sig` changes sign between samples i and i+1 (peaks, valleys, flat edges)"""
    slope_sign = np.sign(np.gradient(sig))
    return np.flatnonzero(slope_sign[1:] != slope_sign[:-1])


def init_liv(x, y, num_sources=1, win_size=9, thres=0.0025, dec=1):
    """Inducing inputs at the extrema of the (Hann-smoothed) signal.

    Returns ([za, zc], u): per source a copy of the (k', 1) extremum locations for the activation and for the
    component GP, and the signal values there.  The envelope threshold only sets HOW MANY extrema are kept: the
    reference applies argsort to the tuple np.where returns (init_models.py:38-43), which enumerates 0..k-1, so the
    result is the FIRST k turning points in time (k = number whose normalised envelope exceeds `thres`), then every
    `dec`-th of those.  Reproduced deliberately: the published 109-point demo depends on it."""
    t = np.asarray(x).reshape(-1)
    s = np.asarray(y).reshape(-1)
    envelope = _hann_average(np.abs(s), 1600)
    envelope = envelope / envelope.max()
    turns = _turning_points(_hann_average(s, win_size))
    k = int(np.count_nonzero(envelope[turns] > thres))
    keep = turns[:k][::dec]
    loc = t[keep].reshape(-1, 1)
    z = [[loc.copy() for _ in range(num_sources)] for _role in ("activation", "component")]
    return z, s[keep].reshape(-1, 1)


def _every_kth_plus_last(x, stride):
    x = np.asarray(x)
    return np.concatenate([x[::stride], x[-1:]], axis=0)


def init_iv(x, num_sources, nivps_a, nivps_c, fs):
    """Uniform inducing inputs: `nivps_a` (`nivps_c`) per second for the activations (components), i.e. every
    (fs // nivps)-th sample — the reference's Python-2 integer division — with the final sample appended (so the
    last point can repeat the grid's last one).  (init_models.py:54-71)"""
    za = [_every_kth_plus_last(x, int(fs // nivps_a)) for _ in range(num_sources)]
    zc = [_every_kth_plus_last(x, int(fs // nivps_c)) for _ in range(num_sources)]
    return [za, zc]


def init_kernel_training(y, list_files, fs, maxh=25):
    """Per training recording: a Matern12(l=1, v=3.5) activation kernel and a Matern32sm component kernel whose
    partial frequencies / variances are the spectral peaks of that recording (methods.init_cparam), held fixed.
    Returns ([kern_act, kern_com], per-recording init_cparam results).  (init_models.py:74-89)"""
    from .kernels import Matern12, Matern32sm
    from .methods import find_ideal_f0, init_cparam
    pitches = find_ideal_f0(list_files)
    spectra = [init_cparam(y[i], fs=fs, maxh=maxh, ideal_f0=pitches[i]) for i in range(len(list_files))]
    kern_act = [Matern12(1, lengthscales=1., variance=3.5) for _ in spectra]
    kern_com = []
    for sp in spectra:
        k = Matern32sm(1, num_partials=len(sp[1]), lengthscales=1., variances=sp[1], frequencies=sp[0])
        k.vars_n_freqs_fixed()
        kern_com.append(k)
    return [kern_act, kern_com], spectra


def _copy_value(p):
    return np.array(p.value, copy=True)


def init_kernel_with_trained_models(m, option_two=False):
    """Kernels of the multi-pitch model from single-pitch models `m` trained beforehand: partial frequencies and
    variances are taken over (variances fixed, frequencies trainable), the envelope hyper-parameters either copied
    from the trained model or, with `option_two`, set to l_act=0.5, v_act=4, l_com=1.  (init_models.py:92-121)"""
    from .kernels import Matern12, Matern32sm
    kern_act, kern_com = [], []
    for trained in m:
        src_act, src_com = trained.kern_act[0], trained.kern_com[0]
        act = Matern12(1)
        com = Matern32sm(1, num_partials=src_com.num_partials)
        com.fixed = True
        com.vars_n_freqs_fixed(fix_var=True, fix_freq=False)
        if option_two:
            act.lengthscales, act.variance, com.lengthscales = 0.5, 4.0, 1.0
        else:
            act.lengthscales = _copy_value(src_act.lengthscales)
            act.variance = _copy_value(src_act.variance)
            com.lengthscales = _copy_value(src_com.lengthscales)
        act.fixed = False
        com.lengthscales.fixed = False
        for j in range(src_com.num_partials):
            com.frequency[j].value = _copy_value(src_com.frequency[j])
            com.variance[j].value = _copy_value(src_com.variance[j])
        kern_act.append(act)
        kern_com.append(com)
    return [kern_act, kern_com]


def get_features(f, s, f_centers, nfpc, use_centers, totalnumf):
    """(frequency, energy) features of a component kernel from a spectrum (f, s).  With `use_centers`: the `nfpc`
    bins around the bin nearest each centre (nfpc // 2 either side; a single bin for nfpc == 1), as (n, 1) columns;
    otherwise the `totalnumf` strongest bins.  Energies are normalised to sum 1.  (init_models.py:154-180)"""
    f, s = np.asarray(f), np.asarray(s)
    if not use_centers:
        strongest = np.argsort(np.log(s))[::-1][:totalnumf]
        energy = s[strongest].copy()
        return f[strongest].copy(), energy / np.sum(energy)
    centres = np.asarray(f_centers).reshape(-1)
    nearest = np.abs(f[None, :] - centres[:, None]).argmin(axis=1)
    lo, hi = (0, 1) if nfpc == 1 else (-(nfpc // 2), nfpc // 2)
    # (the reference slices f[idx + lo : idx + hi]; a centre closer than nfpc // 2 bins to the start gives the same
    # empty or wrapped slice here)
    freq = np.asarray([f[i + lo:i + hi] for i in nearest]).reshape(-1, 1)
    energy = np.asarray([s[i + lo:i + hi] for i in nearest]).reshape(-1, 1)
    return freq, energy / sum(energy)


def init_kern(num_pitches, energy, frequency):
    """Older multi-pitch model: Matern32(l=0.25, v=3.5) activations; components Matern52 x MercerCosMix with the
    Matern52 variance fixed at 1, its lengthscale confined to (0, 0.5) by a Logistic transform, and the cosine
    mixture (variance 0.25, given energies / frequencies) entirely fixed.  (init_models.py:183-198)"""
    from .kernels import Matern32, Matern52, MercerCosMix
    from .param import transforms

    def component(i):
        envelope = Matern52(1, lengthscales=0.25, variance=1.0)
        envelope.variance.fixed = True
        envelope.lengthscales.transform = transforms.Logistic(0., 0.5)
        mixture = MercerCosMix(input_dim=1, energy=np.array(energy[i], copy=True),
                               frequency=np.array(frequency[i], copy=True), variance=0.25, features_as_params=False)
        mixture.fixed = True
        return envelope * mixture
    return [[Matern32(1, lengthscales=0.25, variance=3.5) for _ in range(num_pitches)],
            [component(i) for i in range(num_pitches)]]
