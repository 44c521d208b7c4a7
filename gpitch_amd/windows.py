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


# ----------------------------------------------------------------------------------------------------------------
# device-batched form: W windows per launch sequence (include/gpitch_abi.h gp_sgprb_*), scipy's L-BFGS-B per window
# driven in reverse communication (lbfgsb_batch.py)
class SgprWindowBatch(object):
    """Device side of a batch of W SGPRSS windows that share N, M and the kernel structure of `template` (an SGPRSS):
    bound and gradient of all of them from one launch sequence."""

    def __init__(self, template, num_windows, N, M, handle=None):
        import ctypes as C
        from . import _lib
        self.h = h = handle or template._handle or _lib.default_handle()
        kl = template.kern.kern_list
        P = len(kl)
        i32 = C.c_int32 * P
        self._keep = (i32(*[k.type_code for k in kl]), i32(*[int(k.num_partials) for k in kl]))
        cfg = _lib.SgprConfig(P, int(N), int(M), self._keep[0], self._keep[1], 1e-6, int(bool(template.reg)))
        plan = C.c_void_p()
        h.check(h.lib.gp_sgprb_create(h.h, C.byref(cfg), int(num_windows), C.byref(plan)))
        self.plan = plan
        self.W, self.N, self.M = int(num_windows), int(N), int(M)
        self.nparams = int(h.lib.gp_sgprb_num_params(plan))
        self._ws = h.workspace(h.lib.gp_sgprb_workspace_bytes(plan))
        h.check(h.lib.gp_sgprb_set_workspace(plan, self._ws.data_ptr(), self._ws.numel()))
        t = h.torch
        self.X, self.Y, self.Z = h.zeros(self.W, self.N), h.zeros(self.W, self.N), h.zeros(self.W, self.M)
        self.params, self.grad, self.bound = h.zeros(self.W, self.nparams), h.zeros(self.W, self.nparams), h.zeros(self.W)
        # pinned staging: parameter upload / result download without a pageable-memory synchronisation each round
        self._p_host = t.zeros(self.W, self.nparams, dtype=t.float64).pin_memory()
        self._g_host = t.zeros(self.W, self.nparams, dtype=t.float64).pin_memory()
        self._b_host = t.zeros(self.W, dtype=t.float64).pin_memory()
        self._st_host = t.zeros(4, dtype=t.int32).pin_memory()      # not-positive-definite status of the evaluation in flight
        self._ev, self._pending = None, None
        self.last_status = (0, 0, 0)

    def load(self, xs, ys, zs):
        """put `len(xs)` windows into the first slots (the others keep whatever they held: still valid problems)"""
        t = self.h.torch
        n = len(xs)
        self.X[:n].copy_(t.as_tensor(np.stack([np.asarray(x, dtype=np.float64).reshape(-1) for x in xs])))
        self.Y[:n].copy_(t.as_tensor(np.stack([np.asarray(y, dtype=np.float64).reshape(-1) for y in ys])))
        self.Z[:n].copy_(t.as_tensor(np.stack([np.asarray(z, dtype=np.float64).reshape(-1) for z in zs])))
        self.count = n

    def submit(self, params_host, with_grad=True):
        """enqueue one evaluation (parameter upload, the launch sequence, result download) and return at once; `collect`
        waits for exactly this evaluation — not for whatever else is queued on the stream (another batch's evaluation:
        `fit_windows_batched` keeps two batches in flight so that the host's L-BFGS-B stepping of one overlaps the
        device's evaluation of the other)"""
        h = self.h
        n = self.count
        self._p_host[:n].copy_(h.torch.as_tensor(params_host))
        self.params[:n].copy_(self._p_host[:n], non_blocking=True)
        h.check(h.lib.gp_sgprb_bound_grad(self.plan, self.params.data_ptr(), self.X.data_ptr(), self.Y.data_ptr(),
                                          self.Z.data_ptr(), n, self.bound.data_ptr(),
                                          self.grad.data_ptr() if with_grad else None))
        self._b_host[:n].copy_(self.bound[:n], non_blocking=True)
        if with_grad:
            self._g_host[:n].copy_(self.grad[:n], non_blocking=True)
        # this evaluation's own status word, cleared behind the copy: a failed Cholesky is reported by the evaluation that
        # caused it (status[2] = window slot) and never leaks into the next batch / the next user of the handle
        h.check(h.lib.gp_take_not_pd(h.h, self._st_host.data_ptr()))
        if self._ev is None:
            self._ev = h.torch.cuda.Event()
        self._ev.record(h.torch.cuda.current_stream(h.device))
        self._pending = (n, bool(with_grad))

    def collect(self):
        """(bound, grad) of the evaluation in flight.  `self.last_status` = (flag, pivot, window slot) of the FIRST Cholesky
        failure in it (flag 0: none): that window's numbers — and possibly those of other windows that failed in the same
        evaluation, the device word keeps the first only — are not a bound; `fit_windows_batched` retires the window and
        evaluates again, `evaluate` raises."""
        n, with_grad = self._pending
        self._ev.synchronize()
        self._pending = None
        st = self._st_host.numpy()
        self.last_status = (int(st[0]), int(st[1]), int(st[2]))
        return self._b_host[:n].numpy().copy(), (self._g_host[:n].numpy().copy() if with_grad else None)

    def evaluate(self, params_host, with_grad=True):
        """params_host: (count, nparams) constrained parameter vectors -> (bound (count,), grad (count, nparams))"""
        self.submit(params_host, with_grad)
        out = self.collect()
        if self.last_status[0]:
            from ._lib import GP_ERR_NOT_PD, NotPositiveDefiniteError
            raise NotPositiveDefiniteError(GP_ERR_NOT_PD, "Cholesky failed: window %d of the batch is not positive "
                                           "definite (pivot %d)" % (self.last_status[2], self.last_status[1]))
        return out

    def _load_xnew(self, xnews, n_windows):
        t = self.h.torch
        if xnews is None:
            return self.X[:n_windows], self.N            # predict at the windows' own frames (separation.py:306, 311)
        xn = np.stack([np.asarray(x, dtype=np.float64).reshape(-1) for x in xnews])
        if xn.shape[0] != n_windows:
            raise ValueError("one Xnew per loaded window")
        return t.as_tensor(xn).to(self.X.device), int(xn.shape[1])

    def predict_f(self, params_host, xnews=None):
        """SGPRSS.predict_f of every loaded window at its parameters params_host[i] (constrained vectors, as
        `fit_windows_batched` returns them in "params"): (mean, var), each (count, n) — separation.py:306"""
        h = self.h
        cnt = self.count
        xn, n = self._load_xnew(xnews, cnt)
        if n > self.N:
            raise ValueError("window-batched predict_f takes at most N new points per window")
        self._p_host[:cnt].copy_(h.torch.as_tensor(np.asarray(params_host, dtype=np.float64)))
        self.params[:cnt].copy_(self._p_host[:cnt], non_blocking=True)
        mean, var = h.empty(cnt, n), h.empty(cnt, n)
        h.check(h.lib.gp_sgprb_predict_f(self.plan, self.params.data_ptr(), self.X.data_ptr(), self.Y.data_ptr(),
                                         self.Z.data_ptr(), xn.data_ptr(), n, cnt, mean.data_ptr(), var.data_ptr()))
        return mean.cpu().numpy(), var.cpu().numpy()

    def predict_s(self, params_host, xnews=None, chunk=None):
        """SGPRSS.predict_s (the exact per-source posteriors, sgpr_ss.py:73-114) of every loaded window: (mean, var),
        each (count, P, n) — separation.py:311.  `chunk` windows share one launch sequence (default: as many as fit
        in ~16 GiB of workspace: the N x N factor and its inverse are 64 MB per window at N = 2001)."""
        h = self.h
        cnt = self.count
        xn, n = self._load_xnew(xnews, cnt)
        P = len(self._keep[0])
        self._p_host[:cnt].copy_(h.torch.as_tensor(np.asarray(params_host, dtype=np.float64)))
        self.params[:cnt].copy_(self._p_host[:cnt], non_blocking=True)
        if chunk is None:
            per = max(1, int(h.lib.gp_sgprb_predict_source_workspace_bytes(self.plan, 1, n)))
            chunk = max(1, min(cnt, (16 << 30) // per))
        chunk = int(min(chunk, cnt))
        ws = h.workspace(h.lib.gp_sgprb_predict_source_workspace_bytes(self.plan, chunk, n))
        mean, var = h.empty(cnt, P, n), h.empty(cnt, P, n)
        for b0 in range(0, cnt, chunk):
            c = min(chunk, cnt - b0)
            h.check(h.lib.gp_sgprb_predict_source(self.plan, self.params[b0:].data_ptr(), self.X[b0:].data_ptr(),
                                                  self.Y[b0:].data_ptr(), xn[b0:].data_ptr(), n, c,
                                                  mean[b0:].data_ptr(), var[b0:].data_ptr(), ws.data_ptr(), ws.numel()))
        return mean.cpu().numpy(), var.cpu().numpy()

    def close(self):
        if self.plan is not None:
            self.h.sync()
            self.h.lib.gp_sgprb_destroy(self.plan)
            self.plan = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def fit_windows_batched(make_model, windows, maxiter=10, batch=64, reset=default_reset, handle=None, rank=0,
                        world_size=1, params0=None, predict=False, inflight=2):
    """fit_windows with the device work batched: `batch` windows go through every bound + gradient evaluation together
    (one launch sequence, gp_sgprb_bound_grad), each window driven by its own instance of scipy's L-BFGS-B routine
    (lbfgsb_batch.LbfgsbRC: the iterates of `model.optimize(maxiter=maxiter)` exactly, given the same f and g).

    Every window starts from the parameter values of make_model(handle) after `reset(model, x, y, z)` (unit noise and
    kernel variances by default, transcription.py:253-263) — i.e. carry_kernel_state=False of fit_windows — or from
    params0[i] (constrained vector [noise | theta_0 | ...]) when given.  All windows must share N and M.
    predict=True adds what SoSp.optimize computes after every window's optimisation (separation.py:300-313), batched the
    same way: "mean", "var" (predict_f at the window's frames, (N, 1)) and "smean", "svar" (predict_s: lists over the
    sources of (N, 1) arrays).
    `inflight` batches are kept going at once (default 2: the host's stepping of one overlaps the device's evaluation of
    the other; 1 = one after another).
    `reset` is evaluated ONCE, on this rank's first window, to obtain the starting parameter values of every window (the
    default reset and AMT.reset_model set the same values for every window); per-window starting values go in `params0`.
    The template model must have no mean function and float64 strips (NotImplementedError otherwise).
    A window whose Kuu or B loses positive definiteness during its fit (the reference: tf.cholesky raises out of that
    window's optimize()) — or has them not positive definite at its STARTING parameters already — is retired with
    results[i]["error"] set, bound = nan and its starting parameters; the other windows of the batch are unaffected (their
    evaluation is repeated; the retired slot rides along on a healthy window's parameters, so it cannot fail again).  One
    repeat per failing window: the device's status word names the first failure of an evaluation only.
    Returns a list over windows of dicts: bound, nfev, nit, variances, noise, params [, error]."""
    from . import _lib, lbfgsb_batch
    from .dist import window_assignment
    from .sgpr_ss import SGPRSS
    if not lbfgsb_batch.available():
        raise RuntimeError("fit_windows_batched needs scipy >= 1.15 (its reverse-communication L-BFGS-B routine); "
                           "use fit_windows")
    mine = window_assignment(len(windows), world_size, rank)
    results = [None] * len(windows)
    if not mine:
        return results
    if handle is None:
        # a stream of its own: the launch sequence of an evaluation is recorded into a hipGraph and replayed, which the
        # legacy null stream cannot do
        import torch
        dev0 = _lib.default_handle().device
        s = torch.cuda.Stream(device=dev0)
        with torch.cuda.stream(s):
            hs = _lib.Handle(dev0.index, stream=s)
            try:
                return fit_windows_batched(make_model, windows, maxiter=maxiter, batch=batch, reset=reset, handle=hs,
                                           rank=rank, world_size=world_size, params0=params0, predict=predict,
                                           inflight=inflight)
            finally:
                s.synchronize()
                hs.close()
    h = handle
    model = make_model(h)
    # what the batched plan takes from the template model: kernel structure, `reg`, the parameter values after ONE
    # reset() (on this rank's first window).  It does not carry a mean function or float32 strips: refuse, do not drop.
    if getattr(model, "mean_function", None) is not None and type(model.mean_function).__name__ != "Zero":
        raise NotImplementedError("fit_windows_batched: the template model has a mean_function; the batched plan has "
                                  "none (use fit_windows)")
    if getattr(model, "_bits", 64) != 64:
        raise NotImplementedError("fit_windows_batched runs in float64; the template model asks for float32 strips "
                                  "(use fit_windows)")
    x0w, y0w, z0w = windows[mine[0]][:3]
    reset(model, x0w, y0w, z0w)
    st = model._objective_setup()
    ps, free_idx = st["ps"], st["free_idx"]
    vals0 = st["vals0"]
    N, M = int(np.asarray(x0w).size), int(np.asarray(z0w).size)
    for i in mine:
        if np.asarray(windows[i][0]).size != N or np.asarray(windows[i][2]).size != M:
            raise ValueError("fit_windows_batched: all windows must have the same number of frames and inducing points")
    B = max(1, min(int(batch), len(mine)))
    back = np.array([ps[i].transform.backward(np.atleast_1d(vals0[i]))[0] for i in free_idx])
    var_idx = [int(o) for o in np.cumsum([1] + [2 + 2 * int(k.num_partials) for k in model.kern.kern_list])[:-1]]
    chunks = [mine[b0:b0 + B] for b0 in range(0, len(mine), B)]
    # Two batches in flight, each on a device plan of its own: while the host steps the L-BFGS-B instances of one batch
    # (setulb and the transforms: as long as a device evaluation at W = 256), the device evaluates the other.
    nslots = min(int(inflight), len(chunks))
    devs = [SgprWindowBatch(model, B, N, M, handle=h) for _ in range(max(1, nslots))]

    class _Run(object):
        pass

    def start(dev, ids):
        r = _Run()
        r.dev, r.ids, r.n = dev, ids, len(ids)
        dev.load([windows[i][0] for i in ids], [windows[i][1] for i in ids], [windows[i][2] for i in ids])
        r.base = np.tile(vals0, (r.n, 1))
        if params0 is not None:
            r.base = np.stack([np.asarray(params0[i], dtype=np.float64) for i in ids])
            x0s = [np.array([ps[j].transform.backward(np.atleast_1d(r.base[q, j]))[0] for j in free_idx]) for q in range(r.n)]
        else:
            x0s = [back.copy() for _ in range(r.n)]
        r.runs = [lbfgsb_batch.LbfgsbRC(x0, maxiter=maxiter) for x0 in x0s]
        r.Xf = np.stack(x0s)
        r.active = list(range(r.n))
        r.failed = {}
        submit(r)
        return r

    def submit(r):
        y, r.dy = SGPRSS._free_to_params(st, r.Xf)
        pv = r.base.copy()
        pv[:, free_idx] = y
        # retired windows ride along on the parameters of a healthy window (their own starting parameters may be what failed:
        # Kuu depends on the kernel parameters and Z alone, so a healthy window's parameters factorise in any slot)
        donor = r.active[0] if r.active else None
        for q in r.failed:
            pv[q] = pv[donor] if donor is not None else r.base[q]
        r.dev.submit(pv)

    def advance(r):
        """collect the evaluation in flight, step every active instance; True while the batch still needs evaluations"""
        bound, grad = r.dev.collect()
        flag, pivot, slot = r.dev.last_status
        if flag:
            # TF raises InvalidArgumentError out of the one session.run whose tf.cholesky failed and the reference's window
            # loop dies there (transcription.py:283); here that window alone is retired with the error, and the evaluation
            # is repeated for the others (the device word names the first failure only: another window may have failed in
            # the same evaluation — it is found by the repeat)
            if slot in r.failed or not (0 <= slot < r.n):
                # (a retired slot runs on a healthy window's parameters: it cannot be what failed unless the status word is wrong)
                raise _lib.NotPositiveDefiniteError(_lib.GP_ERR_NOT_PD, "Cholesky failed in window slot %d (pivot %d), which is %s"
                                                    % (slot, pivot, "already retired" if slot in r.failed else "outside the batch"))
            r.failed[slot] = "Cholesky failed: not positive definite (pivot %d) at evaluation %d" % (pivot, r.runs[slot].nfev + 1)
            r.active = [q for q in r.active if q != slot]
            if r.active:
                submit(r)
            return bool(r.active)
        nxt = []
        for q in r.active:
            g = -(grad[q, free_idx] * r.dy[q])
            r.runs[q].give(-bound[q], g)
            if r.runs[q].step():
                r.Xf[q] = r.runs[q].x
                nxt.append(q)
        r.active = nxt
        if nxt:
            submit(r)
        return bool(nxt)

    def finish(r):
        yfin, _ = SGPRSS._free_to_params(st, np.stack([q.x for q in r.runs]))
        pfin = r.base.copy()
        pfin[:, free_idx] = yfin
        for q in r.failed:
            pfin[q] = r.base[q]
        for q, i in enumerate(r.ids):
            results[i] = {"bound": -r.runs[q].fun, "nfev": r.runs[q].nfev, "nit": r.runs[q].nit,
                          "variances": pfin[q, var_idx].copy(), "noise": float(pfin[q, 0]), "params": pfin[q].copy()}
            if q in r.failed:
                results[i].update(error=r.failed[q], bound=float("nan"))
        if predict:
            fm, fv = r.dev.predict_f(pfin)
            sm, sv = r.dev.predict_s(pfin)
            for q, i in enumerate(r.ids):
                if q in r.failed:           # (predicted at the starting parameters only to keep the batch whole)
                    continue
                results[i]["mean"], results[i]["var"] = fm[q].reshape(-1, 1), fv[q].reshape(-1, 1)
                results[i]["smean"] = [sm[q, k].reshape(-1, 1) for k in range(sm.shape[1])]
                results[i]["svar"] = [sv[q, k].reshape(-1, 1) for k in range(sv.shape[1])]

    pending = list(chunks)
    live = []
    for dev in devs:
        if pending:
            live.append(start(dev, pending.pop(0)))
    while live:
        for r in list(live):
            if advance(r):
                continue
            finish(r)
            live.remove(r)
            if pending:
                live.append(start(r.dev, pending.pop(0)))
    for dev in devs:
        dev.close()
    model._destroy()
    return results



def merge_sources(results, ws, n):
    """SoSp.predict_s (separation.py:340-368): overlap-add the per-window source posteriors of
    `fit_windows_batched(..., predict=True)` into whole-signal estimates [[mean_1, var_1], [mean_2, var_2], ...]
    (window_overlap.merged_mean / merged_variance, Hann-weighted halves of 50 %-overlapping windows)."""
    from . import window_overlap
    num_sources = len(results[0]["smean"])
    out = []
    for k in range(num_sources):
        m = window_overlap.merged_mean(y=[r["smean"][k] for r in results], ws=ws, n=n)
        v = window_overlap.merged_variance(y=[r["svar"][k] for r in results], ws=ws, n=n)
        out.append([m, v])
    return out
