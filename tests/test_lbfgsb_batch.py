"""The reverse-communication L-BFGS-B driver (gpitch_amd/lbfgsb_batch.py) against scipy.optimize.minimize itself:
same routine driven the same way, so iterates, function counts and stopping must be IDENTICAL, not merely close."""
import numpy as np
import pytest
from scipy.optimize import minimize

from gpitch_amd import lbfgsb_batch as lb

pytestmark = pytest.mark.skipif(not lb.available(), reason="scipy >= 1.15 private L-BFGS-B routine not present")


def _rosen(x):
    f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g


def _logistic_fit(x, A, b):
    z = A @ x
    f = np.sum(np.logaddexp(0., z) - b * z) + 0.05 * x @ x
    g = A.T @ (1. / (1. + np.exp(-z)) - b) + 0.1 * x
    return f, g


@pytest.mark.parametrize("maxiter", [1, 3, 10, 200])
def test_identical_to_scipy_minimize(maxiter):
    rng = np.random.RandomState(0)
    probs = []
    for k in range(7):
        n = 3 + 5 * k
        if k % 2:
            A, b = rng.randn(40, n), (rng.rand(40) > 0.5).astype(float)
            probs.append((lambda x, A=A, b=b: _logistic_fit(x, A, b), rng.randn(n)))
        else:
            probs.append((_rosen, rng.randn(n) * 0.7))
    ref = [minimize(f, x0, jac=True, method="L-BFGS-B", options=dict(maxiter=maxiter)) for f, x0 in probs]
    runs = [lb.LbfgsbRC(x0, maxiter=maxiter) for _, x0 in probs]
    active = list(range(len(runs)))
    rounds = 0
    while active:                       # every round: one evaluation per still-running problem
        vals = {i: probs[i][0](runs[i].x) for i in active}
        nxt = []
        for i in active:
            runs[i].give(*vals[i])
            if runs[i].step():
                nxt.append(i)
        active = nxt
        rounds += 1
    for r, s in zip(runs, ref):
        np.testing.assert_array_equal(r.x, s.x)
        assert r.fun == s.fun and r.nfev == s.nfev and r.nit == s.nit and r.status == s.status
    assert rounds == max(s.nfev for s in ref)


def test_minimize_many_helper():
    rng = np.random.RandomState(3)
    x0s = [rng.randn(6) for _ in range(5)]

    def batch(X, active):
        out = [_rosen(x) for x in X]
        return np.array([o[0] for o in out]), np.stack([o[1] for o in out])
    runs = lb.minimize_many(batch, x0s, maxiter=10)
    for r, x0 in zip(runs, x0s):
        s = minimize(_rosen, x0, jac=True, method="L-BFGS-B", options=dict(maxiter=10))
        np.testing.assert_array_equal(r.x, s.x)
        assert r.nfev == s.nfev
