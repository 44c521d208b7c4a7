"""scipy's L-BFGS-B routine driven in reverse communication, so that MANY independent minimisations can share one
batched objective evaluation.

GPflow 0.5 `Model.optimize()` (the reference's `model.optimize(disp, maxiter)`, gpitch/transcription.py:283,
gpitch/separation.py:298) is `scipy.optimize.minimize(method='L-BFGS-B', jac=True)` on the free state.  scipy's own
driver loop (`scipy.optimize._lbfgsb_py._minimize_lbfgsb`) calls the compiled routine `setulb`, which hands control
back whenever it needs f and g at a new point.  `LbfgsbRC` below is that loop turned inside out: `step()` runs `setulb`
until an evaluation is needed, the caller evaluates the points of all its instances in one go (one GPU launch
sequence for W windows) and hands the values back with `give()`.  Same routine, same workspace, same stopping rules
and defaults as `minimize(method='L-BFGS-B')`: the iterates are identical (tests/test_lbfgsb_batch.py).

Uses the private module `scipy.optimize._lbfgsb` with the calling convention of scipy 1.15 (checked at import).
"""
import numpy as np

try:
    from scipy.optimize import _lbfgsb
    import scipy
    _SCIPY_OK = tuple(int(v) for v in scipy.__version__.split(".")[:2]) >= (1, 15) and hasattr(_lbfgsb, "setulb")
except Exception:      # pragma: no cover
    _lbfgsb = None
    _SCIPY_OK = False


_SELF_CHECK = [None]


# why available() returned False (None while it has not, or has returned True): for bench / test reports
unavailable_reason = [None]


def _self_check():
    """`setulb` is private: a scipy that keeps the name but changes the argument list or the task codes would fail or,
    worse, iterate differently.  Minimise a 2-D Rosenbrock function through LbfgsbRC and through scipy.optimize.minimize:
    x, fun, nit and nfev must be identical, otherwise this module reports itself unavailable (callers use fit_windows)."""
    from scipy.optimize import minimize

    def fg(x):
        f = 100. * (x[1] - x[0] ** 2) ** 2 + (1. - x[0]) ** 2
        g = np.array([-400. * x[0] * (x[1] - x[0] ** 2) - 2. * (1. - x[0]), 200. * (x[1] - x[0] ** 2)])
        return f, g
    x0 = np.array([-1.2, 1.0])
    try:
        ref = minimize(fg, x0, jac=True, method="L-BFGS-B", options=dict(maxiter=25))
        run = LbfgsbRC(x0, maxiter=25)
        guard = 0
        while run.step() and guard < 1000:
            run.give(*fg(run.x))
            guard += 1
        same = bool(np.array_equal(run.x, ref.x) and run.fun == ref.fun and run.nit == ref.nit and run.nfev == ref.nfev)
        if not same:
            unavailable_reason[0] = ("self-check: iterates differ from scipy.optimize.minimize (x %s, fun %r vs %r, nit %d vs %d, "
                                     "nfev %d vs %d)" % (np.array_equal(run.x, ref.x), run.fun, ref.fun, run.nit, ref.nit, run.nfev, ref.nfev))
        return same
    except Exception as e:
        unavailable_reason[0] = "self-check raised %s: %s" % (type(e).__name__, e)
        return False


def available():
    """True when the installed scipy exposes the reverse-communication routine this module drives AND a small
    self-check reproduces scipy.optimize.minimize's iterates exactly (verified on scipy 1.15.x)"""
    if not _SCIPY_OK:
        if unavailable_reason[0] is None:
            unavailable_reason[0] = "scipy.optimize._lbfgsb.setulb is not importable in this scipy"
        return False
    if _SELF_CHECK[0] is None:
        _SELF_CHECK[0] = _self_check()
        if not _SELF_CHECK[0]:
            import warnings
            import scipy
            warnings.warn("gpitch_amd.lbfgsb_batch is unavailable with scipy %s (%s): window fits fall back to the per-window "
                          "fit_windows path, which is several times slower" % (scipy.__version__, unavailable_reason[0]))
    return _SELF_CHECK[0]


class LbfgsbRC(object):
    """One unconstrained L-BFGS-B minimisation (defaults of scipy.optimize.minimize(method='L-BFGS-B'))."""

    def __init__(self, x0, maxiter=15000, maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000, maxls=20):
        if not _SCIPY_OK:
            raise RuntimeError("scipy.optimize._lbfgsb.setulb (scipy >= 1.15) is not available")
        x0 = np.asarray(x0, dtype=np.float64).ravel()
        n = x0.size
        m = int(maxcor)
        self.n, self.m, self.maxiter, self.maxfun, self.maxls = n, m, int(maxiter), int(maxfun), int(maxls)
        self.factr = ftol / np.finfo(float).eps
        self.pgtol = gtol
        self.x = np.array(x0, dtype=np.float64)
        self.f = np.array(0.0, dtype=np.int32)          # as scipy's driver initialises them
        self.g = np.zeros((n,), dtype=np.int32)
        self.nbd = np.zeros(n, np.int32)
        self.low = np.zeros(n, np.float64)
        self.up = np.zeros(n, np.float64)
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29, dtype=np.float64)
        self.nit = 0
        self.nfev = 0
        self.done = False
        self._xe = None          # last point evaluated, its value and gradient (scipy's ScalarFunction cache)
        self._fe = None
        self._ge = None

    def give(self, f, g):
        """value and gradient at the point `step()` asked for (the current self.x)"""
        self._xe = self.x.copy()
        self._fe = float(f)
        self._ge = np.array(g, dtype=np.float64)
        self.nfev += 1

    def step(self):
        """advance until f, g are needed at self.x (returns True) or the minimisation has ended (returns False)"""
        if self.done:
            return False
        while True:
            if self.task[0] == 3:
                # the routine asked for f, g at self.x: evaluated by now (or cached, as ScalarFunction would)
                if self._xe is None or not np.array_equal(self._xe, self.x):
                    return True
                self.f, self.g = self._fe, self._ge
            self.g = np.asarray(self.g).astype(np.float64)
            _lbfgsb.setulb(self.m, self.x, self.low, self.up, self.nbd, self.f, self.g, self.factr, self.pgtol, self.wa,
                           self.iwa, self.task, self.lsave, self.isave, self.dsave, self.maxls, self.ln_task)
            if self.task[0] == 3:
                continue
            if self.task[0] == 1:
                self.nit += 1
                if self.nit >= self.maxiter:
                    self.task[0] = 5
                    self.task[1] = 504
                elif self.nfev > self.maxfun:
                    self.task[0] = 5
                    self.task[1] = 502
            else:
                self.done = True
                return False

    @property
    def status(self):
        if self.task[0] == 4:
            return 0
        if self.nfev > self.maxfun or self.nit >= self.maxiter:
            return 1
        return 2

    @property
    def fun(self):
        return float(self.f)


def minimize_many(fun_and_grad_batch, x0s, maxiter=15000, **kw):
    """Minimise len(x0s) independent objectives that are evaluated together.

    fun_and_grad_batch(X, active) -> (f, G): X is (W, n) with one point per problem (rows of finished problems hold
    their last point), `active` the indices whose values will be used; f (W,), G (W, n).
    Returns the list of LbfgsbRC states (x, fun, nfev, nit, status)."""
    runs = [LbfgsbRC(x0, maxiter=maxiter, **kw) for x0 in x0s]
    X = np.stack([r.x for r in runs])
    active = list(range(len(runs)))
    # scipy's ScalarFunction evaluates x0 on construction: the first round is that evaluation
    while active:
        f, G = fun_and_grad_batch(X, active)
        nxt = []
        for i in active:
            r = runs[i]
            r.give(f[i], G[i])
            if r.step():
                X[i] = r.x
                nxt.append(i)
        active = nxt
    return runs
