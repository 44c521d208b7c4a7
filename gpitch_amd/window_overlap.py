"""Windowing and Hann overlap-add merging — gpitch/window_overlap.py:7-59,194-220 (host numpy).

These produce / consume the per-window data either side of the hot path: `windowed` (50 % overlap, ws odd)
and `segmented` (non-overlapping) split (x, y) into the independent windows that are the multi-GPU sharding
unit; `merged_mean` / `merged_variance` overlap-add per-window predictions back into one signal.
The reference is Python 2: every `/` below is its integer division, made explicit as `//`.
"""
import numpy as np
from scipy import signal


def _hann(n):
    return signal.windows.hann(n)


def windowed(x, y, ws):
    """window_overlap.py:7-16"""
    n = x.size
    l = (ws - 1) // 2
    nw = (n - ws) // l + 1
    xout, yout = [], []
    for i in range(nw):
        xout.append(x[i * l:i * l + ws].copy().reshape(-1, 1))
        yout.append(y[i * l:i * l + ws].copy().reshape(-1, 1))
    return xout, yout


def _merge(y, ws, n, square):
    nw = len(y)
    ll = (ws - 1) // 2
    y = [yi.copy() for yi in y]          # the reference scales the caller's list in place; we do not
    for i in range(nw):
        win = _hann(ws).reshape(-1, 1)
        if i == 0:
            win[0:ll] = 1.
        elif i == nw - 1:
            win[-ll:] = 1.
        if square:
            win = win ** 2
        y[i] = y[i] * win
    yout = np.zeros((n, 1))
    yout[0:ll] = y[0][0:ll]
    yout[-ll:] = y[-1][-ll:]
    for i in range(nw - 1):
        yout[(i + 1) * ll:(i + 2) * ll + 1] = y[i][ll:].copy() + y[i + 1][0:ll + 1].copy()
    return yout


def merged_mean(y, ws, n):
    """window_overlap.py:19-38: Hann overlap-add of the window means (first/last half-windows kept flat)"""
    return _merge(y, ws, n, False)


def merged_variance(y, ws, n):
    """window_overlap.py:40-59: same with the squared window"""
    return _merge(y, ws, n, True)


def merged_on_device(windows, ws, n, square=False, handle=None):
    """merged_mean (square=False) / merged_variance (square=True) on the GPU: `windows` is a (num_windows, ws) device
    tensor (e.g. the stacked per-window predictions of windows.fit_windows) or anything np.asarray accepts; returns a
    device tensor of n frames.  Same arithmetic as _merge above (gp_overlap_merge, csrc/opt.hip)."""
    from . import _lib
    h = handle or _lib.default_handle()
    t = h.torch
    if not t.is_tensor(windows):
        windows = h.to_device(np.asarray([np.asarray(wi, dtype=np.float64).reshape(-1) for wi in windows]))
    windows = windows.contiguous()
    nw = windows.shape[0]
    out = h.empty(n)
    h.check(h.lib.gp_overlap_merge(h.h, windows.data_ptr(), nw, ws, windows.shape[1], n, int(bool(square)), out.data_ptr()))
    return out


def merged_x(x, ws):
    """window_overlap.py:60-73"""
    l = (ws - 1) // 2
    nw = len(x)
    n = (ws - 1) // 2 * (nw - 1) + ws
    xout = np.zeros((n, 1))
    xout[0:l] = x[0][0:l]
    xout[-l - 1:] = x[-1][-l - 1:]
    for i in range(nw - 1):
        xout[(i + 1) * l:(i + 2) * l] = x[i][-l - 1:-1].copy()
    return xout


def augmentate(x, y, augment_size=1600):
    """window_overlap.py:213-220"""
    addzeros = np.zeros((augment_size, 1))
    yaug1 = np.append(addzeros, y.copy()).reshape(-1, 1)
    yaug = np.append(yaug1, addzeros).reshape(-1, 1)
    alpha = augment_size / 16000.
    xaug = np.linspace(x[0] - alpha, x[-1] + alpha, x.size + 2 * augment_size).reshape(-1, 1)
    return xaug, yaug


def segmented(x, y, window_size=32000, aug=False):
    """window_overlap.py:194-211: non-overlapping segments (trailing remainder dropped)"""
    num_windows = y.size // window_size
    xs, ys = [], []
    for i in range(num_windows):
        yaux = y[i * window_size:(i + 1) * window_size].copy()
        xaux = x[i * window_size:(i + 1) * window_size].copy()
        if aug:
            xaug, yaug = augmentate(xaux, yaux)
        else:
            xaug, yaug = xaux.copy(), yaux.copy()
        ys.append(yaug)
        xs.append(xaug)
    return xs, ys
