"""GPU parity of the C-ABI operators against the oracle (run with -m gpu on an MI355X)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gpflow05 as orc  # noqa: E402  (checker only)


def _theta(k):
    return np.array([k["variance"], k["lengthscales"]] + list(k["energy"]) + list(k["frequency"]), dtype=np.float64)


_KT = {"matern12": 0, "matern32": 1, "matern52": 2, "rbf": 3, "mercer_matern12sm": 4, "matern12sm": 5,
       "matern32sm": 6, "mercer_matern52sm": 7}


def _desc(h, k):
    from gpitch_amd import _lib
    th = h.to_device(_theta(k))
    d = _lib.KernelDesc(_KT[k["type"]], len(k["frequency"]), th.data_ptr())
    return d, th


KERNELS = [
    {"type": "matern12", "variance": 1.3, "lengthscales": 0.7, "energy": [], "frequency": []},
    {"type": "matern32", "variance": 3.5, "lengthscales": 1.0, "energy": [], "frequency": []},
    {"type": "matern52", "variance": 0.8, "lengthscales": 0.3, "energy": [], "frequency": []},
    {"type": "rbf", "variance": 2.0, "lengthscales": 0.25, "energy": [], "frequency": []},
    {"type": "mercer_matern12sm", "variance": 1.1, "lengthscales": 0.1, "energy": [0.5, 0.3, 0.2],
     "frequency": [261.6, 523.2, 784.9]},
    {"type": "mercer_matern12sm", "variance": 1.0, "lengthscales": 0.1, "energy": [1. / 20] * 20,
     "frequency": [261.6 * (i + 1) for i in range(20)]},
    {"type": "matern12sm", "variance": 0.9, "lengthscales": 0.2, "energy": [0.6, 0.4], "frequency": [100., 205.]},
    {"type": "matern32sm", "variance": 1.0, "lengthscales": 0.7, "energy": [0.12, 0.2, 0.05],
     "frequency": [110., 221., 330.5]},
    {"type": "mercer_matern52sm", "variance": 0.25, "lengthscales": 0.25, "energy": [0.5, 0.3, 0.1, 0.1],
     "frequency": [196., 392., 588., 784.]},
]


@pytest.mark.parametrize("kern", KERNELS, ids=lambda k: "%s_m%d" % (k["type"], len(k["frequency"])))
@pytest.mark.parametrize("n1,n2", [(7, 13), (64, 513), (109, 1000)])
def test_kernel_build_matches_oracle(gp_handle, kern, n1, n2):
    h = gp_handle
    rng = np.random.RandomState(n1 * 1000 + n2)
    x2 = np.sort(rng.rand(n2, 1), 0) * 0.5
    x1 = x2[rng.choice(n2, n1, replace=False)].copy() if n1 <= n2 else np.sort(rng.rand(n1, 1), 0)
    d, th = _desc(h, kern)
    dx1, dx2 = h.to_device(x1), h.to_device(x2)
    out = h.empty(n1, n2)
    h.check(h.lib.gp_kernel_build(h.h, C.byref(d), dx1.data_ptr(), n1, dx2.data_ptr(), n2, out.data_ptr(), n2, 0))
    ref = orc.K(kern, x1, x2)
    np.testing.assert_allclose(out.cpu().numpy(), ref, rtol=1e-11, atol=1e-12)
    # K(X) form and accumulate
    outs = h.empty(n1, n1)
    h.check(h.lib.gp_kernel_build(h.h, C.byref(d), dx1.data_ptr(), n1, None, 0, outs.data_ptr(), n1, 0))
    np.testing.assert_allclose(outs.cpu().numpy(), orc.K(kern, x1, None), rtol=1e-11, atol=1e-12)
    h.check(h.lib.gp_kernel_build(h.h, C.byref(d), dx1.data_ptr(), n1, None, 0, outs.data_ptr(), n1, 1))
    np.testing.assert_allclose(outs.cpu().numpy(), 2 * orc.K(kern, x1, None), rtol=1e-11, atol=1e-12)
    kd = h.empty(n2)
    h.check(h.lib.gp_kernel_diag(h.h, C.byref(d), n2, kd.data_ptr(), 0))
    np.testing.assert_allclose(kd.cpu().numpy(), orc.Kdiag(kern, x2), rtol=1e-15)


@pytest.mark.parametrize("M", [5, 32, 33, 64, 109, 256, 512])
def test_kuu_cholesky_and_inverse(gp_handle, M):
    h = gp_handle
    kern = KERNELS[1]
    z = np.linspace(0, 2.0, M).reshape(-1, 1)
    d, th = _desc(h, kern)
    dz = h.to_device(z)
    L, W = h.empty(M, M), h.empty(M, M)
    ws = h.workspace(h.lib.gp_chol_workspace_bytes(M))
    h.check(h.lib.gp_kuu_cholesky(h.h, C.byref(d), dz.data_ptr(), M, 1e-6, L.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel()))
    Kuu = orc.K(kern, z, None) + 1e-6 * np.eye(M)
    Lref = np.linalg.cholesky(Kuu)
    Lg, Wg = L.cpu().numpy(), W.cpu().numpy()
    assert np.all(np.triu(Lg, 1) == 0) and np.all(np.triu(Wg, 1) == 0)
    # backward-stable factorisation: L L^T reproduces Kuu to rounding
    np.testing.assert_allclose(Lg @ Lg.T, Kuu, rtol=0, atol=1e-12 * np.abs(Kuu).max() * M)
    np.testing.assert_allclose(Lg, Lref, rtol=0, atol=1e-7 * np.abs(Lref).max())
    np.testing.assert_allclose(Wg @ Lg, np.eye(M), rtol=0, atol=1e-8)


@pytest.mark.parametrize("M,ld", [(75, 75), (300, 302), (512, 512), (620, 620), (1100, 1100)])
def test_cholesky_inplace_one_workgroup(gp_handle, M, ld):
    """gp_cholesky_inplace = chol_kernel on one matrix: whole 32-row tiles take the lean update (C block loaded into the
    accumulators, negated-operand MFMA), ragged sizes the general tile; from M = 620 on the panel no longer fits the LDS and
    the operands come from the matrix itself; 1100 rows also leave the tile table (1056 rows) for the index arithmetic"""
    h = gp_handle
    rng = np.random.RandomState(M)
    z = np.sort(rng.rand(M)) * 2.0
    K = np.exp(-np.abs(z[:, None] - z[None, :]) / 0.3) + 1e-6 * np.eye(M)
    A = np.zeros((M, ld))
    A[:, :M] = K
    dA = h.to_device(A)
    h.check(h.lib.gp_cholesky_inplace(h.h, dA.data_ptr(), M, ld))
    L = dA.cpu().numpy()[:, :M]
    assert np.all(np.triu(L, 1) == 0)
    np.testing.assert_allclose(L @ L.T, K, rtol=0, atol=1e-12 * M)
    np.testing.assert_allclose(L, np.linalg.cholesky(K), rtol=0, atol=1e-8)


def test_cholesky_reports_not_pd(gp_handle):
    from gpitch_amd import _lib
    h = gp_handle
    A = np.eye(40)
    A[17, 17] = -1.0
    dA = h.to_device(A)
    st = h.lib.gp_cholesky_inplace(h.h, dA.data_ptr(), 40, 40)
    assert st == _lib.GP_ERR_NOT_PD
    assert h.lib.gp_last_not_pd_index(h.h) == 17
    with pytest.raises(_lib.NotPositiveDefiniteError):
        h.check(st)


@pytest.mark.parametrize("whiten", [True, False])
@pytest.mark.parametrize("kern", [KERNELS[1], KERNELS[4], KERNELS[5]], ids=["matern32", "mercer3", "mercer20"])
@pytest.mark.parametrize("N,M", [(50, 7), (1000, 109), (4096, 64), (3001, 256)])
def test_conditional_matches_oracle(gp_handle, kern, whiten, N, M):
    h = gp_handle
    rng = np.random.RandomState(N + M)
    x = np.linspace(0, (N - 1) / 16000., N).reshape(-1, 1)
    z = x[:: max(N // M, 1)][:M].copy()
    M = z.shape[0]
    q_mu = 0.3 * rng.randn(M, 1)
    q_sqrt = (np.eye(M) + 0.05 * rng.randn(M, M))[:, :, None]   # full matrix: upper part must be ignored
    d, th = _desc(h, kern)
    dx, dz, dmu, dsq = h.to_device(x), h.to_device(z), h.to_device(q_mu), h.to_device(q_sqrt[:, :, 0])
    fm, fv = h.empty(N), h.empty(N)
    ws = h.workspace(h.lib.gp_conditional_workspace_bytes(N, M))
    h.check(h.lib.gp_conditional_diag(h.h, C.byref(d), dx.data_ptr(), N, dz.data_ptr(), M, dmu.data_ptr(), dsq.data_ptr(),
                                      int(whiten), 1e-6, fm.data_ptr(), fv.data_ptr(), ws.data_ptr(), ws.numel()))
    rm, rv = orc.conditional(x, z, kern, q_mu, q_sqrt, whiten)
    scale = np.abs(rm).max() + 1e-300
    # these inducing grids are far denser than the lengthscale: cond(Kuu) ~ 1e9 (jitter-limited), and the
    # unwhitened form applies Kuu^-1 (not just L^-1), so both sides carry ~cond*eps error there.
    tol = 2e-8 if whiten else 2e-6
    np.testing.assert_allclose(fm.cpu().numpy(), rm[:, 0], rtol=0, atol=tol * scale)
    np.testing.assert_allclose(fv.cpu().numpy(), rv[:, 0], rtol=0, atol=tol * np.abs(rv).max())


@pytest.mark.parametrize("nlin", [0, 1, 2])
@pytest.mark.parametrize("P", [1, 3, 12])
def test_mpd_varexp_matches_oracle(gp_handle, nlin, P):
    h = gp_handle
    N = 1537
    rng = np.random.RandomState(P * 10 + nlin)
    Fmu = rng.randn(N, 2 * P) * 2.0 + 1.0
    Fvar = rng.rand(N, 2 * P) * 3.0 + 1e-8
    y = rng.randn(N, 1)
    nv = h.to_device(np.array([0.37]))
    pf = h.empty(N)
    s = C.c_double()
    dmu, dvar, dy = h.to_device(Fmu), h.to_device(Fvar), h.to_device(y)   # keep alive across the call
    h.check(h.lib.gp_mpd_varexp(h.h, dmu.data_ptr(), dvar.data_ptr(), dy.data_ptr(), N, P, nlin,
                                nv.data_ptr(), pf.data_ptr(), C.byref(s)))
    ref = orc.mpd_variational_expectations(Fmu, Fvar, y, 0.37, P, nlin)
    np.testing.assert_allclose(pf.cpu().numpy(), ref[:, 0], rtol=1e-11, atol=1e-11)
    assert abs(s.value - ref.sum()) <= 1e-11 * abs(ref.sum())


@pytest.mark.parametrize("ktype", ["matern32", "mercer_matern12sm"])
def test_gauss_kl_with_prior_covariance(gp_handle, ktype):
    """gauss_kl(q_mu, q_sqrt, K) with K = kern.K(z) + jitter I (pdgp.py:126-129): through the kernel descriptor
    (gp_gauss_kl), through an explicit matrix (gp_gauss_kl_matrix) and through the Python mirror."""
    from gpitch_amd.conditionals import gauss_kl
    h = gp_handle
    M = 45
    rng = np.random.RandomState(5)
    z = (np.linspace(0, 0.05, M) + 2e-4 * rng.rand(M)).reshape(-1, 1)
    q_mu = rng.randn(M, 1)
    q_sqrt = (np.eye(M) + 0.1 * rng.randn(M, M))[:, :, None]
    if ktype == "matern32":
        kd = {"type": "matern32", "variance": 1.3, "lengthscales": 0.02, "energy": [], "frequency": []}
    else:
        kd = {"type": "mercer_matern12sm", "variance": 0.9, "lengthscales": 0.05, "energy": [0.5, 0.3, 0.2],
              "frequency": [110.0, 220.0, 330.0]}
    K = orc.K(kd, z) + 1e-6 * np.eye(M)
    ref = float(orc.gauss_kl(q_mu, q_sqrt, K))
    desc, keep = _desc(h, kd)
    out = C.c_double()
    ws = h.workspace(h.lib.gp_gauss_kl_workspace_bytes(M, 1))
    dmu, dsq, dz = h.to_device(q_mu), h.to_device(q_sqrt[:, :, 0]), h.to_device(z)
    h.check(h.lib.gp_gauss_kl(h.h, dmu.data_ptr(), dsq.data_ptr(), M, C.byref(desc), dz.data_ptr(), 1e-6,
                              C.byref(out), ws.data_ptr(), ws.numel()))
    assert abs(out.value - ref) <= 1e-8 * abs(ref), (out.value, ref)
    dK = h.to_device(K)
    out2 = C.c_double()
    h.check(h.lib.gp_gauss_kl_matrix(h.h, dmu.data_ptr(), dsq.data_ptr(), M, dK.data_ptr(), C.byref(out2),
                                     ws.data_ptr(), ws.numel()))
    assert abs(out2.value - ref) <= 1e-8 * abs(ref)
    assert abs(gauss_kl(q_mu, q_sqrt, K) - ref) <= 1e-8 * abs(ref)
    # not positive definite -> status, not a wrong number
    bad = K.copy(); bad[3, 3] = -1.0
    dB = h.to_device(bad)
    with pytest.raises(Exception):
        h.check(h.lib.gp_gauss_kl_matrix(h.h, dmu.data_ptr(), dsq.data_ptr(), M, dB.data_ptr(), C.byref(out2),
                                         ws.data_ptr(), ws.numel()))


def test_gauss_kl_whitened(gp_handle):
    h = gp_handle
    M = 77
    rng = np.random.RandomState(3)
    q_mu = rng.randn(M, 1)
    q_sqrt = (np.eye(M) + 0.1 * rng.randn(M, M))[:, :, None]
    out = C.c_double()
    ws = h.workspace(8192)
    dmu, dsq = h.to_device(q_mu), h.to_device(q_sqrt[:, :, 0])
    h.check(h.lib.gp_gauss_kl(h.h, dmu.data_ptr(), dsq.data_ptr(), M, None, None, 1e-6,
                              C.byref(out), ws.data_ptr(), ws.numel()))
    ref = orc.gauss_kl(q_mu, q_sqrt)
    assert abs(out.value - ref) <= 1e-12 * abs(ref)


def test_logistic_transform_and_adam_step(gp_handle):
    """gpflow.transforms.Logistic(a, b) (kernels.py:219-223, init_models.py:189) on the device: forward / backward
    round trip, and one Adam step on a mixed free-state vector against the host formulas."""
    from gpitch_amd.param import Identity, Log1pe, Logistic
    h = gp_handle
    tr = [Identity(), Log1pe(), Logistic(0., 2.), Logistic(0., 0.25), Logistic(0., 0.5), Logistic(-1., 3.)]
    codes = [t.device_code(h) for t in tr]
    assert codes[:2] == [0, 1] and len(set(codes)) == len(codes) and all(c >= 3 for c in codes[2:])
    assert Logistic(0., 2.).device_code(h) == codes[2]            # registering the same pair again reuses its code
    rng = np.random.RandomState(0)
    n = 600
    which = rng.randint(0, len(tr), n)
    x = rng.randn(n) * 2.0
    tc = np.array([codes[w] for w in which], dtype=np.uint8)
    y_ref = np.array([tr[w].forward(np.array([xi]))[0] for w, xi in zip(which, x)])
    t = h.torch
    dx, dtc, dy, dback = h.to_device(x), t.as_tensor(tc, device=h.device), h.empty(n), h.empty(n)
    h.check(h.lib.gp_transform_forward(h.h, dx.data_ptr(), dtc.data_ptr(), n, dy.data_ptr()))
    np.testing.assert_allclose(dy.cpu().numpy(), y_ref, rtol=1e-14, atol=1e-15)
    h.check(h.lib.gp_transform_backward(h.h, dy.data_ptr(), dtc.data_ptr(), n, dback.data_ptr()))
    np.testing.assert_allclose(dback.cpu().numpy(), x, rtol=1e-8, atol=1e-8)
    # one Adam step (TF-1.2 rule) maximising: g_free = -(grad * dy/dx)
    grad = rng.randn(n)
    m, v = h.zeros(n), h.zeros(n)
    dg = h.to_device(grad)
    lr, b1, b2, eps = 0.01, 0.9, 0.999, 1e-8
    h.check(h.lib.gp_adam_step(h.h, dx.data_ptr(), dy.data_ptr(), dg.data_ptr(), dtc.data_ptr(), m.data_ptr(),
                               v.data_ptr(), n, 1, lr, b1, b2, eps))
    gf = -grad * np.array([tr[w].dforward(np.array([xi]))[0] for w, xi in zip(which, x)])
    m1, v1 = (1 - b1) * gf, (1 - b2) * gf * gf
    x1 = x - lr * np.sqrt(1 - b2) / (1 - b1) * m1 / (np.sqrt(v1) + eps)
    np.testing.assert_allclose(dx.cpu().numpy(), x1, rtol=1e-12, atol=1e-14)
    y1 = np.array([tr[w].forward(np.array([xi]))[0] for w, xi in zip(which, x1)])
    np.testing.assert_allclose(dy.cpu().numpy(), y1, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("ws,nw", [(2001, 5), (31, 2), (101, 17)])
def test_overlap_merge_on_device_matches_host(gp_handle, ws, nw):
    """gp_overlap_merge = window_overlap.merged_mean / merged_variance (window_overlap.py:19-59), boundary frames and
    flat half-windows included."""
    from gpitch_amd.window_overlap import merged_mean, merged_variance, merged_on_device
    rng = np.random.RandomState(ws + nw)
    ll = (ws - 1) // 2
    n = ll * (nw + 1) + 1
    ys = [rng.randn(ws, 1) for _ in range(nw)]
    for square, host in ((False, merged_mean), (True, merged_variance)):
        ref = host([y.copy() for y in ys], ws, n).reshape(-1)
        got = merged_on_device(ys, ws, n, square=square, handle=gp_handle).cpu().numpy()
        np.testing.assert_allclose(got, ref, rtol=1e-13, atol=1e-14)
    bad = gp_handle.empty(nw, ws)
    with pytest.raises(Exception):
        gp_handle.check(gp_handle.lib.gp_overlap_merge(gp_handle.h, bad.data_ptr(), nw, ws, ws, n + 1, 0, bad.data_ptr()))


@pytest.mark.parametrize("M", [40, 200])
def test_kuu_cholesky_with_inverse_reports_not_pd(gp_handle, M):
    """the one-launch factor + inverse (LDS copy for M <= 64, global otherwise) reports a bad pivot like the plain
    factorisation: a kernel with a negative variance gives -K, not positive definite at the first pivot"""
    from gpitch_amd import _lib
    h = gp_handle
    kern = dict(KERNELS[1])
    kern["variance"] = -1.0
    z = np.linspace(0, 2.0, M).reshape(-1, 1)
    d, th = _desc(h, kern)
    dz = h.to_device(z)
    L, W = h.empty(M, M), h.empty(M, M)
    ws = h.workspace(h.lib.gp_chol_workspace_bytes(M))
    st = h.lib.gp_kuu_cholesky(h.h, C.byref(d), dz.data_ptr(), M, 1e-6, L.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel())
    assert st == _lib.GP_ERR_NOT_PD
    assert h.lib.gp_last_not_pd_index(h.h) == 0


@pytest.mark.parametrize("whiten", [True, False])
@pytest.mark.parametrize("kern", [KERNELS[1], KERNELS[4]], ids=["matern32", "mercer3"])
@pytest.mark.parametrize("N,M,with_q", [(50, 7, True), (333, 64, True), (200, 130, False)])
def test_conditional_full_cov_matches_oracle(gp_handle, kern, whiten, N, M, with_q):
    """full_cov=True (GPflow 0.5 conditionals.conditional): N x N x 1 covariance against the oracle"""
    from gpitch_amd.conditionals import conditional
    from gpitch_amd.synth import kernels_from_problem
    rng = np.random.RandomState(N + M)
    x = np.linspace(0, (N - 1) / 16000., N).reshape(-1, 1)
    z = x[:: max(N // M, 1)][:M].copy()
    M = z.shape[0]
    q_mu = 0.3 * rng.randn(M, 1)
    q_sqrt = (np.eye(M) + 0.05 * rng.randn(M, M))[:, :, None] if with_q else None
    k = kernels_from_problem({"kern_act": [kern], "kern_com": []})[0][0]
    fm, fc = conditional(x, z, k, q_mu, full_cov=True, q_sqrt=q_sqrt, whiten=whiten)
    rm, rc = orc.conditional(x, z, kern, q_mu, q_sqrt, whiten, full_cov=True)
    assert fc.shape == (N, N, 1)
    tol = 2e-8 if whiten else 2e-6
    np.testing.assert_allclose(fm, rm, rtol=0, atol=tol * (np.abs(rm).max() + 1e-300))
    np.testing.assert_allclose(fc, rc, rtol=0, atol=tol * np.abs(rc).max())
    _, fv = conditional(x, z, k, q_mu, q_sqrt=q_sqrt, whiten=whiten)
    # K(x, x) carries exp(-r(x, x)) with r(x, x) = sqrt(1e-12) (euclid_dist) where Kdiag is exact: the diagonals agree
    # to 1e-6 of the prior variance and no closer — in the reference too
    np.testing.assert_allclose(np.diag(fc[:, :, 0]), fv[:, 0], rtol=0, atol=3e-6 * max(np.abs(fv).max(), kern["variance"]))


@pytest.mark.parametrize("M,ld", [(128, 128), (160, 164), (352, 352), (512, 520)])
def test_cholesky_by_a_workgroup_cluster(gp_handle, M, ld):
    """chol_cluster.hip (one matrix, 128 <= M <= 512, M % 32 == 0): factor in place with and without the inverse, three
    launches in a row on the same buffers (the exchange flags carry the launch's epoch: nothing is cleared in between),
    a row stride wider than the matrix, and a bad pivot in the middle of the chain"""
    from gpitch_amd import _lib
    h = gp_handle
    rng = np.random.RandomState(M)
    B = rng.randn(M, M)
    K = B @ B.T / M + np.diag(0.5 + rng.rand(M))
    Lref = np.linalg.cholesky(K)
    A = np.zeros((M, ld))
    for rep in range(3):
        A[:, :M] = K * (1.0 + rep)
        dA = h.to_device(A)
        h.check(h.lib.gp_cholesky_inplace(h.h, dA.data_ptr(), M, ld))
        L = dA.cpu().numpy()[:, :M]
        assert np.all(np.triu(L, 1) == 0)
        np.testing.assert_allclose(L, Lref * np.sqrt(1.0 + rep), rtol=0, atol=1e-12 * np.abs(Lref).max() * M)
    if ld == M:         # the factor + inverse entry point builds its own Kuu: Matern-3/2 on a grid (cond ~ 1e6)
        kern = KERNELS[1]
        z = np.linspace(0, 2.0, M).reshape(-1, 1)
        d, th = _desc(h, kern)
        dz = h.to_device(z)
        Ld, Wd = h.empty(M, M), h.empty(M, M)
        ws = h.workspace(h.lib.gp_chol_workspace_bytes(M))
        for rep in range(2):
            h.check(h.lib.gp_kuu_cholesky(h.h, C.byref(d), dz.data_ptr(), M, 1e-6, Ld.data_ptr(), Wd.data_ptr(), ws.data_ptr(), ws.numel()))
        Kuu = orc.K(kern, z, None) + 1e-6 * np.eye(M)
        Lg, Wg = Ld.cpu().numpy(), Wd.cpu().numpy()
        assert np.all(np.triu(Lg, 1) == 0) and np.all(np.triu(Wg, 1) == 0)
        np.testing.assert_allclose(Lg @ Lg.T, Kuu, rtol=0, atol=1e-12 * np.abs(Kuu).max() * M)
        np.testing.assert_allclose(Wg @ Lg, np.eye(M), rtol=0, atol=1e-8)
    Kbad = K.copy()
    Kbad[M // 2 + 3, M // 2 + 3] = -5.0
    A[:, :M] = Kbad
    dA = h.to_device(A)
    st = h.lib.gp_cholesky_inplace(h.h, dA.data_ptr(), M, ld)
    assert st == _lib.GP_ERR_NOT_PD
    assert h.lib.gp_last_not_pd_index(h.h) == M // 2 + 3
    A[:, :M] = K        # and the next launch on that buffer is clean again
    dA.copy_(h.torch.as_tensor(A))
    h.check(h.lib.gp_cholesky_inplace(h.h, dA.data_ptr(), M, ld))
    np.testing.assert_allclose(dA.cpu().numpy()[:, :M], Lref, rtol=0, atol=1e-12 * np.abs(Lref).max() * M)


def test_concurrent_clusters_on_separate_handles():
    """four host threads, each with its own handle and stream, factorising (and inverting) 512 x 512 matrices at the same
    time: the clusters' exchange areas are per (handle, matrix), their workgroups share the device"""
    import threading
    import torch
    from gpitch_amd import _lib
    M, nthreads, reps = 512, 4, 6
    kern = KERNELS[1]
    z = np.linspace(0, 2.0, M).reshape(-1, 1)
    Kuu = orc.K(kern, z, None) + 1e-6 * np.eye(M)
    Lref = np.linalg.cholesky(Kuu)
    errs = [None] * nthreads

    def work(t):
        try:
            dev = _lib.default_handle().device
            stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream):
                h = _lib.Handle(dev.index, stream=stream)
                d, th = _desc(h, kern)
                dz = h.to_device(z)
                L, W = h.empty(M, M), h.empty(M, M)
                ws = h.workspace(h.lib.gp_chol_workspace_bytes(M))
                worst = 0.0
                for _ in range(reps):
                    h.check(h.lib.gp_kuu_cholesky(h.h, C.byref(d), dz.data_ptr(), M, 1e-6, L.data_ptr(), W.data_ptr(), ws.data_ptr(), ws.numel()))
                    Lg, Wg = L.cpu().numpy(), W.cpu().numpy()
                    worst = max(worst, np.abs(Lg - Lref).max() / np.abs(Lref).max(), np.abs(Wg @ Lg - np.eye(M)).max())
                errs[t] = worst
        except Exception as e:      # noqa: BLE001
            errs[t] = e

    ths = [threading.Thread(target=work, args=(t,)) for t in range(nthreads)]
    for th_ in ths:
        th_.start()
    for th_ in ths:
        th_.join()
    for e in errs:
        assert not isinstance(e, Exception), e
        assert e is not None and e <= 1e-7, errs


def test_cluster_factorisation_at_every_tile_count(gp_handle):
    """every size the workgroup cluster takes (M = 128, 160, .. 512: 4 .. 16 tiles a side, 2 .. 17 workgroups): factor and
    inverse of a Matern-3/2 Kuu against numpy"""
    h = gp_handle
    kern = KERNELS[1]
    for M in range(128, 513, 32):
        z = np.linspace(0, 2.0, M).reshape(-1, 1)
        d, th = _desc(h, kern)
        dz = h.to_device(z)
        Ld, Wd = h.empty(M, M), h.empty(M, M)
        ws = h.workspace(h.lib.gp_chol_workspace_bytes(M))
        h.check(h.lib.gp_kuu_cholesky(h.h, C.byref(d), dz.data_ptr(), M, 1e-6, Ld.data_ptr(), Wd.data_ptr(), ws.data_ptr(), ws.numel()))
        Kuu = orc.K(kern, z, None) + 1e-6 * np.eye(M)
        Lg, Wg = Ld.cpu().numpy(), Wd.cpu().numpy()
        assert np.all(np.triu(Lg, 1) == 0) and np.all(np.triu(Wg, 1) == 0), M
        np.testing.assert_allclose(Lg @ Lg.T, Kuu, rtol=0, atol=1e-12 * np.abs(Kuu).max() * M, err_msg=str(M))
        np.testing.assert_allclose(Wg @ Lg, np.eye(M), rtol=0, atol=1e-8, err_msg=str(M))


def test_cluster_exchange_areas_are_capped(gp_handle):
    """150 different matrix buffers factorised through the cluster on one handle: the exchange areas (one per matrix address)
    are released in halves past 64, results stay right before and after"""
    h = gp_handle
    M = 128
    rng = np.random.RandomState(1)
    B = rng.randn(M, M)
    K = B @ B.T / M + np.eye(M)
    Lref = np.linalg.cholesky(K)
    bufs = []
    for rep in range(150):
        dA = h.to_device(K)
        bufs.append(dA)              # (kept alive: every buffer is a new address)
        h.check(h.lib.gp_cholesky_inplace(h.h, dA.data_ptr(), M, M))
        if rep % 37 == 0 or rep == 149:
            np.testing.assert_allclose(dA.cpu().numpy(), Lref, rtol=0, atol=1e-12 * M)
    h.check(h.lib.gp_cholesky_inplace(h.h, bufs[0].copy_(h.torch.as_tensor(K)).data_ptr(), M, M))     # an evicted address again
    np.testing.assert_allclose(bufs[0].cpu().numpy(), Lref, rtol=0, atol=1e-12 * M)
