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


def init_kernel_training(y, list_files, fs, maxh=25):
    """init_models.py:74-89: one (Matern12 activation, Matern32sm component) pair per training file, the
    component initialised from the note's spectrum."""
    from .kernels import Matern12, Matern32sm
    from .methods import find_ideal_f0, init_cparam
    num_pitches = len(list_files)
    if0 = find_ideal_f0(list_files)
    iparam, kern_act, kern_com = [], [], []
    for i in range(num_pitches):
        iparam.append(init_cparam(y[i], fs=fs, maxh=maxh, ideal_f0=if0[i]))
        kern_act.append(Matern12(1, lengthscales=1., variance=3.5))
        kern_com.append(Matern32sm(1, num_partials=len(iparam[i][1]), lengthscales=1., variances=iparam[i][1],
                                   frequencies=iparam[i][0]))
        kern_com[i].vars_n_freqs_fixed()
    return [kern_act, kern_com], iparam


def init_kernel_with_trained_models(m, option_two=False):
    """init_models.py:92-121: kernels of the multi-pitch model from the per-pitch trained models `m`."""
    from .kernels import Matern12, Matern32sm
    kern_act, kern_com = [], []
    num_sources = len(m)
    for i in range(num_sources):
        num_p = m[i].kern_com[0].num_partials
        kern_act.append(Matern12(1))
        kern_com.append(Matern32sm(1, num_partials=num_p))
        kern_act[i].fixed = True
        kern_com[i].fixed = True
        kern_com[i].vars_n_freqs_fixed(fix_var=True, fix_freq=False)
        if option_two:
            kern_act[i].lengthscales = 0.5
            kern_act[i].variance = 4.0
            kern_com[i].lengthscales = 1.0
        else:
            kern_act[i].lengthscales = m[i].kern_act[0].lengthscales.value.copy()
            kern_act[i].variance = m[i].kern_act[0].variance.value.copy()
            kern_com[i].lengthscales = m[i].kern_com[0].lengthscales.value.copy()
        kern_act[i].fixed = False
        kern_com[i].lengthscales.fixed = False
        for j in range(num_p):
            kern_com[i].frequency[j].value = m[i].kern_com[0].frequency[j].value.copy()
            kern_com[i].variance[j].value = m[i].kern_com[0].variance[j].value.copy()
    return [kern_act, kern_com]


def get_features(f, s, f_centers, nfpc, use_centers, totalnumf):
    """Get kernel features (parameters) from FFT of training data (init_models.py:154-180)"""
    if use_centers:
        var_l, freq_l = [], []
        for i in range(f_centers.size):
            idx = np.argmin(np.abs(f - f_centers[i]))
            if nfpc == 1:
                freq_l.append(f[idx: idx + 1])
                var_l.append(s[idx: idx + 1])
            else:
                freq_l.append(f[idx - nfpc // 2: idx + nfpc // 2])
                var_l.append(s[idx - nfpc // 2: idx + nfpc // 2])
        frequency = np.asarray(freq_l).reshape(-1, 1)
        energy = np.asarray(var_l).reshape(-1, 1)
        energy = energy / sum(energy)
    else:
        num_features = totalnumf
        idx = np.flip(np.argsort(np.log(s)), axis=0)
        ssorted = s[idx].copy()
        fsorted = f[idx].copy()
        energy = ssorted[0:num_features].copy()
        energy /= np.sum(energy)
        frequency = fsorted[0:num_features].copy()
    return frequency, energy


def init_kern(num_pitches, energy, frequency):
    """Initialize kernels for activations and components (init_models.py:183-198): Matern32 activations and
    Matern52 * MercerCosMix components (Matern52 variance fixed, lengthscale ~ Logistic(0, 0.5); the cosine
    mixture fixed)."""
    from .kernels import Matern32, Matern52, MercerCosMix
    from .param import transforms
    k_act, k_com = [], []
    for i in range(num_pitches):
        k_act.append(Matern32(1, lengthscales=0.25, variance=3.5))
        a = Matern52(1, lengthscales=0.25, variance=1.0)
        a.variance.fixed = True
        a.lengthscales.transform = transforms.Logistic(0., 0.5)
        b = MercerCosMix(input_dim=1, energy=np.asarray(energy[i]).copy(), frequency=np.asarray(frequency[i]).copy(),
                         variance=0.25, features_as_params=False)
        b.fixed = True
        k_com.append(a * b)
    return [k_act, k_com]
