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


def _half(ws):
    return (ws - 1) // 2


def windowed(x, y, ws):
    """Half-overlapping windows of `ws` samples (hop (ws - 1) // 2; whatever does not fill a last window is
    dropped), as two lists of (ws, 1) arrays.  (window_overlap.py:7-16)"""
    hop = _half(ws)
    count = (np.size(x) - ws) // hop + 1
    if count <= 0:
        return [], []
    starts = hop * np.arange(count)

    def cut(a):
        frames = np.lib.stride_tricks.sliding_window_view(np.asarray(a).reshape(-1), ws)[starts]
        return [f.reshape(-1, 1).copy() for f in frames]
    return cut(x), cut(y)


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
    """The time axis the windows of `windowed` came from, rebuilt from the windows: first half-window of the first
    one, the second hop of every window but the last, the tail of the last.  (window_overlap.py:60-73)"""
    hop = _half(ws)
    stack = np.stack([np.asarray(xi).reshape(-1) for xi in x])
    count = stack.shape[0]
    out = np.zeros(hop * (count - 1) + ws)
    out[:hop] = stack[0, :hop]
    out[out.size - hop - 1:] = stack[-1, -hop - 1:]
    out[hop:count * hop] = stack[:-1, -hop - 1:-1].reshape(-1)
    return out.reshape(-1, 1)


def augmentate(x, y, augment_size=1600):
    """`augment_size` zeros before and after `y`; the time axis extended by augment_size / 16000 s either side over
    the new length (the reference hard-codes 16 kHz here).  (window_overlap.py:213-220)"""
    yaug = np.pad(np.asarray(y, dtype=np.float64).reshape(-1), augment_size).reshape(-1, 1)
    margin = augment_size / 16000.
    xaug = np.linspace(x[0] - margin, x[-1] + margin, np.size(x) + 2 * augment_size).reshape(-1, 1)
    return xaug, yaug


def segmented(x, y, window_size=32000, aug=False):
    """Consecutive non-overlapping segments of `window_size` samples (a trailing remainder is dropped), optionally
    zero-padded by `augmentate`.  (window_overlap.py:194-211)"""
    count = np.size(y) // window_size

    def cut(a):
        a = np.asarray(a)
        blocks = a[:count * window_size].reshape((count, window_size) + a.shape[1:])
        return [b.copy() for b in blocks]
    xs, ys = cut(x), cut(y)
    if aug:
        pairs = [augmentate(xi, yi) for xi, yi in zip(xs, ys)]
        xs, ys = [p[0] for p in pairs], [p[1] for p in pairs]
    return xs, ys
