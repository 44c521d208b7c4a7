"""Many-window SGPRSS fitting — the loop of AMT.optimize / SoSp.optimize (gpitch/transcription.py:265-288,
gpitch/separation.py:279-313) with the windows of one GPU spread over several HIP streams.

The reference fits its windows (ws = 2001 frames, 50 % overlap: window_overlap.py:7-16) one after another; each
L-BFGS-B evaluation is a few dozen small kernels, so one window cannot fill an MI355X.  Here `num_streams` worker
threads each own a HIP stream, a library handle and ONE model whose engine plan and workspace are reused for every
window they draw (the data holders are swapped, as transcription.py:253-263 does); kernels of different windows
overlap on the device.  Across GPUs the windows are dealt round-robin (dist.window_assignment); there is no
exchange between windows.
"""
import threading

import numpy as np


def default_reset(model, x, y, z):
    """AMT.reset_model (transcription.py:253-263): new data, unit noise and kernel variances.  (The reference
    also scales y by 20 there and resets the lengthscales from its params file: callers do that in `reset`.)"""
    model.X = x
    model.Y = y
    model.Z = z
    model.likelihood.variance = 1.
    for k in model.kern.kern_list:
        k.variance = 1.


def fit_windows(make_model, windows, maxiter=10, num_streams=4, reset=default_reset, after_fit=None,
                method='L-BFGS-B', rank=0, world_size=1, carry_kernel_state=False):
    """Fit every window (x_i, y_i, z_i) in `windows` with its own L-BFGS-B run.

    carry_kernel_state=False: every window starts from the parameter values the model was built with, so the
    result does not depend on which worker fitted which window.  The reference never resets the energies and
    frequencies of its kernels between windows (they are trainable, matern12_spectral_mixture.py:86-94, and
    reset_model only touches variances / lengthscales), so there each window starts from the previous window's
    optimum: carry_kernel_state=True with num_streams=1 reproduces that sequential drift.

    make_model(handle) -> SGPRSS   built once per worker (any window's data; it is replaced by `reset`)
    reset(model, x, y, z)          puts a window into the model (default: AMT.reset_model)
    after_fit(model, index) -> obj what to keep per window (default: bound, nfev and the kernel variances, the
                                   piano-roll entries of transcription.py:286-288)
    Returns a list over windows (None for windows owned by other ranks)."""
    import torch
    from . import _lib
    from .dist import window_assignment
    mine = window_assignment(len(windows), world_size, rank)
    results = [None] * len(windows)
    lock = threading.Lock()
    cursor = [0]
    errors = []
    dev = _lib.default_handle().device      # also loads the library / binds the device in the main thread

    def keep(model, idx, res):
        if after_fit is not None:
            return after_fit(model, idx)
        return {"bound": -float(res.fun), "nfev": int(res.nfev),
                "variances": np.array([k.variance.value[0] for k in model.kern.kern_list]),
                "noise": float(model.likelihood.variance.value[0])}

    def worker():
        try:
            torch.cuda.set_device(dev)
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                h = _lib.Handle(dev.index, stream=s)
                model = make_model(h)
                params0 = [p.value.copy() for p in model._param_list()]
                while True:
                    with lock:
                        if cursor[0] >= len(mine):
                            break
                        idx = mine[cursor[0]]
                        cursor[0] += 1
                    x, y, z = windows[idx][:3]
                    if not carry_kernel_state:
                        for p, v in zip(model._param_list(), params0):
                            p.value = v
                    reset(model, x, y, z)
                    res = model.optimize(method=method, maxiter=maxiter)
                    results[idx] = keep(model, idx, res)
                model._destroy()
                s.synchronize()
                h.close()
        except Exception as e:     # surfaced to the caller below
            errors.append(e)

    n = max(1, min(int(num_streams), len(mine) or 1))
    threads = [threading.Thread(target=worker) for _ in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return results
