"""Analytic known-answer tests that pin the oracle.  CPU only.
The reference has no tests of its own; these identities are what stands in for them."""
import numpy as np
import pytest
from scipy.linalg import cho_factor, cho_solve

from oracle import gpflow05 as orc

RNG = np.random.RandomState(0)


def _mercer(m=3, f0=220.):
    return {"type": "mercer_matern12sm", "variance": 1.3, "lengthscales": 0.1, "energy": list(RNG.rand(m) + 0.1),
            "frequency": [f0 * (k + 1) for k in range(m)]}


M32 = {"type": "matern32", "variance": 3.5, "lengthscales": 1.0, "energy": [], "frequency": []}


def test_K2_mercer_feature_identity_and_matern12sm_agreement():
    k = _mercer(4)
    x = np.sort(RNG.rand(40, 1), 0) * 0.05
    z = np.sort(RNG.rand(9, 1), 0) * 0.05
    phi, phi2 = orc.phi_features(k, z), orc.phi_features(k, x)
    direct = sum(e * np.cos(2 * np.pi * f * (z - x.T)) for e, f in zip(k["energy"], k["frequency"]))
    np.testing.assert_allclose(phi.T @ phi2, direct, atol=1e-12)
    kb = dict(k, type="matern12sm")
    np.testing.assert_allclose(orc.K(k, z, x), orc.K(kb, z, x), atol=5e-5)   # differ only in the 1e-12 placement
    np.testing.assert_allclose(orc.Kdiag(k, x), k["variance"] * sum(k["energy"]), rtol=1e-15)


def test_Kdiag_is_exact_not_diag_of_K():
    # r(x,x) = 1e-6, so diag(K(X)) carries exp(-1e-6); Kdiag does not (m12sm.py:119-121)
    k = _mercer(2)
    x = RNG.rand(5, 1)
    assert np.all(np.abs(np.diag(orc.K(k, x)) / orc.Kdiag(k, x) - np.exp(-1e-6)) < 1e-12)


def test_K4_whitened_unwhitened_conditional_and_kl_consistency():
    for kern in (M32, _mercer(3)):
        M, N = 8, 30
        z = np.linspace(0, 2, M).reshape(-1, 1)
        x = np.sort(RNG.rand(N, 1), 0) * 2
        L = np.linalg.cholesky(orc.K(kern, z) + 1e-6 * np.eye(M))
        qmu_w = RNG.randn(M, 1)
        qs_w = np.tril(np.eye(M) + 0.1 * RNG.randn(M, M))
        mw, vw = orc.conditional(x, z, kern, qmu_w, qs_w[:, :, None], whiten=True)
        mu, vu = orc.conditional(x, z, kern, L @ qmu_w, (L @ qs_w)[:, :, None], whiten=False)
        np.testing.assert_allclose(mw, mu, atol=1e-8)
        np.testing.assert_allclose(vw, vu, atol=1e-8)
        klw = orc.gauss_kl(qmu_w, qs_w[:, :, None])
        klu = orc.gauss_kl(L @ qmu_w, (L @ qs_w)[:, :, None], orc.K(kern, z) + 1e-6 * np.eye(M))
        assert abs(klw - klu) < 1e-8 * max(1, abs(klw))
        assert klw >= 0


def test_K1_prior_elbo_identity():
    P, N, M = 2, 200, 11
    x = np.linspace(0, 0.1, N).reshape(-1, 1)
    y = RNG.randn(N, 1)
    ka = [dict(M32, variance=3.5), dict(M32, variance=2.0)]
    kc = [_mercer(3), _mercer(2, 330.)]
    z = [x[::N // M][:M].copy() for _ in range(P)]
    qmu = [np.zeros((M, 1))] * P
    qs = [np.eye(M)[:, :, None]] * P
    s2 = 0.3
    elbo = orc.pdgp_elbo(x, y, z, z, ka, kc, qmu, qs, qmu, qs, s2)
    tot = np.sum(y ** 2)
    for i in range(P):
        _, E2 = orc.hermgauss1d(np.zeros((1, 1)), np.full((1, 1), ka[i]["variance"]), 20, orc.logistic)
        tot += N * kc[i]["variance"] * sum(kc[i]["energy"]) * E2[0, 0]
    ref = -0.5 * tot / s2 - 0.5 * N * (np.log(2 * np.pi) + np.log(s2))
    assert abs(elbo - ref) < 1e-9 * abs(ref)


def test_K3_sgpr_bound_tight_when_Z_equals_X():
    N = 25
    X = np.sort(RNG.rand(N, 1), 0)
    Y = RNG.randn(N, 1)
    kl = [_mercer(2), dict(_mercer(2, 300.), lengthscales=0.3)]
    s2 = 0.2
    b = orc.sgpr_bound(X, Y, X, kl, s2)
    Kxx = orc.K_sum(kl, X) + s2 * np.eye(N)
    c = cho_factor(Kxx, lower=True)
    lml = -0.5 * Y.T @ cho_solve(c, Y) - np.sum(np.log(np.diag(c[0]))) - 0.5 * N * np.log(2 * np.pi)
    # the bound is computed with the jitter-inflated Kuu but the exact Kdiag: equal up to ~jitter * N / sigma^2
    assert b <= lml[0, 0] + 1e-9
    assert abs(b - lml[0, 0]) < 5e-4
    # and a bound with fewer inducing points is looser
    assert orc.sgpr_bound(X, Y, X[::3], kl, s2) <= b + 1e-9


def test_K5_exact_posterior_limit_sgpr_predict():
    N = 20
    X = np.sort(RNG.rand(N, 1), 0)
    Y = np.sin(6 * X) + 0.1 * RNG.randn(N, 1)
    kl = [dict(M32, variance=1.0, lengthscales=0.3)]
    s2 = 0.01
    Xs = np.linspace(0, 1, 7).reshape(-1, 1)
    mean, var = orc.sgpr_predict_f(Xs, X, Y, X, kl, s2)
    Kxx = orc.K_sum(kl, X) + s2 * np.eye(N)
    Ks = orc.K_sum(kl, X, Xs)
    ref_mean = Ks.T @ np.linalg.solve(Kxx, Y)
    ref_var = orc.Kdiag_sum(kl, Xs) - np.sum(Ks * np.linalg.solve(Kxx, Ks), 0)
    np.testing.assert_allclose(mean, ref_mean, atol=2e-4)
    np.testing.assert_allclose(var[:, 0], ref_var, atol=2e-4)
    ms, vs = orc.sgpr_predict_source(Xs, X, Y, kl, s2)
    np.testing.assert_allclose(ms[0], ref_mean, atol=1e-9)


def test_K6_gauss_hermite_against_mpmath():
    import mpmath as mp
    mp.mp.dps = 30
    for mu, s2 in [(-6., 1e-8), (0., 1.0), (3.14, 0.5), (5., 4.0), (12., 9.0)]:
        E1, E2 = orc.hermgauss1d(np.array([[mu]]), np.array([[s2]]), 20, orc.logistic)
        sig = lambda g: 1 / (1 + mp.exp(-2 * (g - mp.pi)))
        pdf = lambda g: mp.exp(-(g - mu) ** 2 / (2 * s2)) / mp.sqrt(2 * mp.pi * s2)
        lo, hi = mu - 12 * np.sqrt(s2), mu + 12 * np.sqrt(s2)
        e1 = mp.quad(lambda g: sig(g) * pdf(g), [lo, mu, hi])
        e2 = mp.quad(lambda g: sig(g) ** 2 * pdf(g), [lo, mu, hi])
        # a 20-point rule is an approximation: it must be close, and essentially exact for narrow q
        # (the reference's rule is what is restated; its own truncation error grows with the width of q)
        tol = 1e-12 if s2 < 1e-6 else (2e-4 if s2 <= 1.0 else 5e-3)
        assert abs(E1[0, 0] - float(e1)) < tol and abs(E2[0, 0] - float(e2)) < tol


def test_K7_cross_term_identity():
    P, N = 5, 50
    E1 = [RNG.rand(N, 1) for _ in range(P)]
    E2 = [RNG.rand(N, 1) for _ in range(P)]
    mf = [RNG.randn(N, 1) for _ in range(P)]
    vf = [RNG.rand(N, 1) for _ in range(P)]
    Y = RNG.randn(N, 1)
    ve = orc.log_lik_exp(Y, mf, vf, E1, E2, 0.5, P)
    a = [E1[i] * mf[i] for i in range(P)]
    A = sum(a)
    B = sum(E2[i] * (vf[i] + mf[i] ** 2) for i in range(P))
    C = A ** 2 - sum(ai ** 2 for ai in a)
    ref = -0.5 * ((Y ** 2 - 2 * Y * A + B + C) / 0.5 + np.log(2 * np.pi) + np.log(0.5))
    np.testing.assert_allclose(ve, ref, atol=1e-12)


def test_K9_notebook_anchor_midi2freq():
    from gpitch_amd.synth import midi2freq
    assert midi2freq(60) == 261.6255653005986   # demos/notebooks/demo_modgp-real-audio.ipynb:66


def test_K10_properties():
    k = _mercer(3)
    x = np.sort(RNG.rand(30, 1), 0) * 0.1
    Kxx = orc.K(k, x)
    np.testing.assert_allclose(Kxx, Kxx.T, atol=1e-14)
    assert np.linalg.eigvalsh(Kxx + 1e-6 * np.eye(30)).min() > 0
    z = x[::4].copy()
    M = z.shape[0]
    m, v = orc.conditional(x, z, k, RNG.randn(M, 1), np.tril(RNG.randn(M, M))[:, :, None], whiten=True)
    assert np.all(v > -1e-9)


def test_transforms_and_minibatch_semantics():
    x = np.array([-30., -1., 0., 2., 40.])
    y = orc.positive_forward(x)
    assert np.all(y > 0)
    np.testing.assert_allclose(orc.positive_backward(y)[1:], x[1:], rtol=1e-9)
    rng = np.random.RandomState(0)
    idx = orc.minibatch_indices(rng, 100, 10)
    assert idx.shape == (10,) and idx.max() < 100
    idx = orc.minibatch_indices(np.random.RandomState(0), 50, 50)
    assert sorted(idx) == list(range(50))
