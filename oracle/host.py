"""CPU restatement of the reference's host-side producers/consumers either side of the hot path
(TEST INFRASTRUCTURE ONLY — imported by tests/ alone, never by gpitch_amd/).

Plain loops, one statement per step of the reference, Python-2 integer division written `//`:
  init_liv       gpitch/init_models.py:9-51       (incl. the argsort-of-a-tuple step at :38-43)
  init_iv        gpitch/init_models.py:54-71
  cparam_select  gpitch/methods.py:117-153        (what init_cparam does AFTER the peak picker)
  peak_pick      PeakUtils 1.x `indexes` (third-party dependency called at gpitch/methods.py:115; MIT licence; not in
                 the reference tree) restated from its published description
  windowed / merged_x / segmented / augmentate   gpitch/window_overlap.py:7-16, 60-73, 194-220
Pinned by the only anchor the reference prints for this code: init_liv(win_size=31, thres=0.033, dec=9) on
demos/data/011PFNOF_M60_train.wav gives 109 points (demos/notebooks/demo_modgp-real-audio.ipynb:88,116;
tests/golden/init_liv_real_audio.npz).  Everything else here is PARITY UNPINNED.
"""
import numpy as np
from scipy import signal


def init_liv(x, y, num_sources=1, win_size=9, thres=0.0025, dec=1):
    x = np.asarray(x).reshape(-1)
    y = np.asarray(y).reshape(-1)
    w_env = signal.windows.hann(1600)                                       # :19
    env = signal.convolve(np.abs(y), w_env, mode='same') / sum(w_env)       # :20
    env = env / np.max(env)                                                 # :21
    w_s = signal.windows.hann(win_size)                                     # :24
    ys = signal.convolve(y, w_s, mode='same') / sum(w_s)                    # :25
    sgn = np.sign(np.gradient(ys))                                          # :28
    where_change = np.where(np.diff(sgn))                                   # :29-30 (a 1-tuple)
    xa, ya, ea = x[where_change], y[where_change], env[where_change]        # :33-35
    above = np.where(ea > thres)                                            # :38 (a 1-tuple)
    order = np.argsort(above)                                               # :41 argsort of the TUPLE -> [[0..k-1]]
    xf = xa[order].reshape(-1, 1)                                           # :42
    yf = ya[order].reshape(-1, 1)                                           # :43
    za, zc = [], []
    for _ in range(num_sources):                                            # :47-49
        za.append(xf[::dec].copy())
        zc.append(xf[::dec].copy())
    return [za, zc], yf[::dec]


def init_iv(x, num_sources, nivps_a, nivps_c, fs):
    da, dc = fs // nivps_a, fs // nivps_c                                   # :65-66 (py2 int division)
    za, zc = [], []
    for _ in range(num_sources):
        za.append(np.vstack([x[::da], x[-1]]))                              # :68
        zc.append(np.vstack([x[::dc], x[-1]]))                              # :69
    return [za, zc]


def peak_pick(y, thres, min_dist):
    """PeakUtils `indexes`: sequential form (plateaus one at a time, suppression with a 'removed' mask)."""
    y = np.asarray(y, dtype=float)
    level = thres * (np.max(y) - np.min(y)) + np.min(y)
    min_dist = int(min_dist)
    dy = np.diff(y)
    n = dy.size
    if not np.any(dy != 0):
        return np.array([], dtype=int)
    runs, i = [], 0
    while i < n:                                  # maximal runs of zero slope
        if dy[i] == 0:
            j = i
            while j + 1 < n and dy[j + 1] == 0:
                j += 1
            runs.append((i, j))
            i = j + 1
        else:
            i += 1
    for (a, b) in runs:
        if a == 0:
            dy[a:b + 1] = dy[b + 1]
        elif b == n - 1:
            dy[a:b + 1] = dy[a - 1]
        else:
            med = np.median(np.arange(a, b + 1))
            left, right = dy[a - 1], dy[b + 1]
            for k in range(a, b + 1):
                dy[k] = left if k < med else right
    peaks = []
    for i in range(y.size):
        d_out = dy[i] if i < n else 0.
        d_in = dy[i - 1] if i > 0 else 0.
        if d_out < 0. and d_in > 0. and y[i] > level:
            peaks.append(i)
    peaks = np.array(peaks, dtype=int)
    if peaks.size > 1 and min_dist > 1:
        removed = np.ones(y.size, dtype=bool)
        removed[peaks] = False
        for p in peaks[np.argsort(y[peaks])][::-1]:
            if not removed[p]:
                removed[max(0, p - min_dist):p + min_dist + 1] = True
                removed[p] = False
        peaks = np.arange(y.size)[~removed]
    return peaks


def cparam_select(F, S, idx, maxh, ideal_f0, scaled=True):
    """methods.py:117-153 given the peak indices `idx`: returns (freq_final, var_final)."""
    Fs, Ss = F[idx], S[idx]
    o = np.argsort(Fs)
    Ss, Fs = Ss[o], np.sort(Fs)
    F2 = S2 = None
    for index in range(Fs.size):                                            # :124-130 (result of the LAST pass wins)
        if Fs[index] < 0.75 * ideal_f0:
            F2, S2 = np.delete(Fs, [index]), np.delete(Ss, [index])
        else:
            F2, S2 = Fs.copy(), Ss.copy()
    a1 = np.flip(np.sort(S2), 0)
    a2 = np.flip(np.argsort(S2), 0)
    if a1.size > maxh:
        a1, a2 = a1[:maxh], a2[:maxh]
    if scaled:
        a1 = a1 * (1. / np.sum(a1))
    ff = F2[a2]
    o = np.argsort(ff)
    return np.sort(ff), a1[o]


def windowed(x, y, ws):
    n = x.size
    l = (ws - 1) // 2
    nw = (n - ws) // l + 1
    xo, yo = [], []
    for i in range(nw):
        xo.append(x[i * l:i * l + ws].copy().reshape(-1, 1))
        yo.append(y[i * l:i * l + ws].copy().reshape(-1, 1))
    return xo, yo


def merged_x(x, ws):
    l = (ws - 1) // 2
    nw = len(x)
    out = np.zeros(((ws - 1) // 2 * (nw - 1) + ws, 1))
    out[0:l] = x[0][0:l]
    out[-l - 1:] = x[-1][-l - 1:]
    for i in range(nw - 1):
        out[(i + 1) * l:(i + 2) * l] = x[i][-l - 1:-1]
    return out


def augmentate(x, y, augment_size=1600):
    z = np.zeros((augment_size, 1))
    ya = np.append(np.append(z, y), z).reshape(-1, 1)
    alpha = augment_size / 16000.
    xa = np.linspace(x[0] - alpha, x[-1] + alpha, x.size + 2 * augment_size).reshape(-1, 1)
    return xa, ya


def segmented(x, y, window_size=32000, aug=False):
    xs, ys = [], []
    for i in range(y.size // window_size):
        xi = x[i * window_size:(i + 1) * window_size].copy()
        yi = y[i * window_size:(i + 1) * window_size].copy()
        if aug:
            xi, yi = augmentate(xi, yi)
        xs.append(xi)
        ys.append(yi)
    return xs, ys
