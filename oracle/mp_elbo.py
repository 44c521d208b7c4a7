"""50-digit mpmath evaluation of the Pdgp ELBO and SGPR bound (TEST INFRASTRUCTURE ONLY).

Independent high-precision evaluation of the SAME formulas restated in gpflow05.py
(reference: gpitch/pdgp.py:113-170, gpitch/likelihoods.py:33-68, gpitch/sgpr_ss.py:29-71,
GPflow-0.5 conditional/gauss_kl).  It pins the oracle's *numerics* (known-answer test K8 of
SURVEY.md §8c); it cannot pin its *semantics* — parity with the reference stays unpinned.

The Gauss-Hermite nodes/weights are the float64 values numpy returns (they are data of the
reference's algorithm, likelihoods.py:35), converted exactly.
"""
import numpy as np
import mpmath as mp

mp.mp.dps = 50


def _m(a):
    a = np.asarray(a, dtype=np.float64)
    return mp.matrix(a.tolist()) if a.ndim == 2 else [mp.mpf(float(v)) for v in a.ravel()]


def _chol(A):
    return mp.cholesky(A)


def _trsm_lower(L, B):
    n, k = B.rows, B.cols
    X = mp.matrix(n, k)
    for c in range(k):
        for i in range(n):
            s = B[i, c]
            for j in range(i):
                s -= L[i, j] * X[j, c]
            X[i, c] = s / L[i, i]
    return X


def _trsm_upper(U, B):
    n, k = B.rows, B.cols
    X = mp.matrix(n, k)
    for c in range(k):
        for i in reversed(range(n)):
            s = B[i, c]
            for j in range(i + 1, n):
                s -= U[i, j] * X[j, c]
            X[i, c] = s / U[i, i]
    return X


def kern_K(kern, x1, x2):
    """x1, x2: lists of mpf.  r = sqrt(((x-x')/l)^2 + 1e-12) for the stationary kernels."""
    v = mp.mpf(float(kern["variance"]))
    t = kern["type"]
    out = mp.matrix(len(x1), len(x2))
    eps = mp.mpf("1e-12")
    for i, a in enumerate(x1):
        for j, b in enumerate(x2):
            if t == "matern12sm":
                r = mp.sqrt((a - b + eps) ** 2)
                s = mp.mpf(0)
                for e, f in zip(kern["energy"], kern["frequency"]):
                    s += mp.mpf(float(e)) * mp.cos(2 * mp.pi * mp.mpf(float(f)) * r)
                out[i, j] = v * mp.exp(-r / mp.mpf(float(kern["lengthscales"]))) * s
                continue
            l = mp.mpf(float(kern["lengthscales"]))
            r2 = ((a - b) / l) ** 2
            r = mp.sqrt(r2 + eps)
            if t == "mercer_matern12sm":
                s = mp.mpf(0)
                for e, f in zip(kern["energy"], kern["frequency"]):
                    s += mp.mpf(float(e)) * mp.cos(2 * mp.pi * mp.mpf(float(f)) * (a - b))
                out[i, j] = v * mp.exp(-r) * s
            elif t == "matern12":
                out[i, j] = v * mp.exp(-r)
            elif t == "matern32":
                out[i, j] = v * (1 + mp.sqrt(3) * r) * mp.exp(-mp.sqrt(3) * r)
            elif t == "matern52":
                out[i, j] = v * (1 + mp.sqrt(5) * r + mp.mpf(5) / 3 * r * r) * mp.exp(-mp.sqrt(5) * r)
            elif t == "rbf":
                out[i, j] = v * mp.exp(-r2 / 2)
            else:
                raise ValueError(t)
    return out



def kern_Kdiag(kern):
    v = mp.mpf(float(kern["variance"]))
    if kern["type"] in ("mercer_matern12sm", "matern12sm"):
        return v * sum(mp.mpf(float(e)) for e in kern["energy"])
    return v


def conditional(xnew, z, kern, q_mu, q_sqrt, whiten):
    M = len(z)
    Kmn = kern_K(kern, z, xnew)
    Kmm = kern_K(kern, z, z)
    for i in range(M):
        Kmm[i, i] += mp.mpf("1e-6")
    Lm = _chol(Kmm)
    A = _trsm_lower(Lm, Kmn)
    N = len(xnew)
    kd = kern_Kdiag(kern)
    fvar = [kd - sum(A[m, n] ** 2 for m in range(M)) for n in range(N)]
    if not whiten:
        A = _trsm_upper(Lm.T, A)
    f = _m(np.asarray(q_mu).reshape(-1))
    fmean = [sum(A[m, n] * f[m] for m in range(M)) for n in range(N)]
    Lq = mp.matrix(np.tril(np.asarray(q_sqrt)[:, :, 0]).tolist())
    LTA = Lq.T * A
    fvar = [fvar[n] + sum(LTA[m, n] ** 2 for m in range(M)) for n in range(N)]
    return fmean, fvar


def gauss_kl(q_mu, q_sqrt, Kp=None):
    M = np.asarray(q_mu).size
    Lq = mp.matrix(np.tril(np.asarray(q_sqrt)[:, :, 0]).tolist())
    mu = mp.matrix([[v] for v in _m(np.asarray(q_mu).reshape(-1))])
    if Kp is None:
        alpha = mu
        trace = sum(Lq[i, j] ** 2 for i in range(M) for j in range(M))
    else:
        Lp = _chol(Kp)
        alpha = _trsm_lower(Lp, mu)
        T = _trsm_lower(Lp, Lq)
        trace = sum(T[i, j] ** 2 for i in range(M) for j in range(M))
    twoKL = sum(alpha[i, 0] ** 2 for i in range(M)) - M - sum(mp.log(Lq[i, i] ** 2) for i in range(M)) + trace
    if Kp is not None:
        twoKL += sum(mp.log(Lp[i, i] ** 2) for i in range(M))
    return twoKL / 2


def _nlin(code, x):
    if code == 0:
        return 1 / (1 + mp.exp(-2 * (x - mp.pi)))
    if code == 1:
        return mp.log(mp.exp(x) + 1)
    return mp.exp(-2 * (x - mp.pi) ** 2)


def hermgauss1d(mean_g, var_g, nlin_code, H=20):
    gx, gw = np.polynomial.hermite.hermgauss(H)
    gx = [mp.mpf(float(v)) for v in gx]
    gw = [mp.mpf(float(v)) for v in (gw / np.sqrt(np.pi))]
    E1, E2 = [], []
    for mg, vg in zip(mean_g, var_g):
        s = mp.sqrt(2 * vg)
        e1 = mp.mpf(0)
        e2 = mp.mpf(0)
        for xh, wh in zip(gx, gw):
            ev = _nlin(nlin_code, xh * s + mg)
            e1 += ev * wh
            e2 += ev * ev * wh
        E1.append(e1)
        E2.append(e2)
    return E1, E2


def pdgp_elbo(x, y, za, zc, kern_act, kern_com, q_mu_act, q_sqrt_act, q_mu_com, q_sqrt_com,
              noise_var, num_data=None, whiten=True, nlin_code=0, return_parts=False):
    P = len(kern_act)
    xs = _m(np.asarray(x).reshape(-1))
    ys = _m(np.asarray(y).reshape(-1))
    N = len(xs)
    s2 = mp.mpf(float(noise_var))
    kl = mp.mpf(0)
    mg, vg, mf, vf = [], [], [], []
    for i in range(P):
        zai = _m(np.asarray(za[i]).reshape(-1))
        zci = _m(np.asarray(zc[i]).reshape(-1))
        m_, v_ = conditional(xs, zai, kern_act[i], q_mu_act[i], q_sqrt_act[i], whiten)
        mg.append(m_); vg.append(v_)
        m_, v_ = conditional(xs, zci, kern_com[i], q_mu_com[i], q_sqrt_com[i], whiten)
        mf.append(m_); vf.append(v_)
        if whiten:
            kl += gauss_kl(q_mu_act[i], q_sqrt_act[i]) + gauss_kl(q_mu_com[i], q_sqrt_com[i])
        else:
            Ka = kern_K(kern_act[i], zai, zai)
            Kc = kern_K(kern_com[i], zci, zci)
            for j in range(len(zai)):
                Ka[j, j] += mp.mpf("1e-6")
            for j in range(len(zci)):
                Kc[j, j] += mp.mpf("1e-6")
            kl += gauss_kl(q_mu_act[i], q_sqrt_act[i], Ka) + gauss_kl(q_mu_com[i], q_sqrt_com[i], Kc)
    E1, E2 = [], []
    for i in range(P):
        e1, e2 = hermgauss1d(mg[i], vg[i], nlin_code)
        E1.append(e1); E2.append(e2)
    total = mp.mpf(0)
    for n in range(N):
        a = [E1[i][n] * mf[i][n] for i in range(P)]
        A = sum(a)
        B = sum(E2[i][n] * (vf[i][n] + mf[i][n] ** 2) for i in range(P))
        C = 2 * sum(a[i] * a[j] for i in range(P - 1) for j in range(i + 1, P))
        total += -(((ys[n] ** 2 - 2 * ys[n] * A + B + C) / s2) + mp.log(2 * mp.pi) + mp.log(s2)) / 2
    scale = mp.mpf(float(N if num_data is None else num_data)) / N
    elbo = total * scale - kl
    if return_parts:
        fmean = np.array([[float(v) for v in col] for col in (mg + mf)]).T
        fvar = np.array([[float(v) for v in col] for col in (vg + vf)]).T
        return float(elbo), float(kl), fmean, fvar
    return float(elbo)


def sgpr_bound(X, Y, Z, kern_list, noise_var):
    xs = _m(np.asarray(X).reshape(-1))
    zs = _m(np.asarray(Z).reshape(-1))
    ys = _m(np.asarray(Y).reshape(-1))
    N, M = len(xs), len(zs)
    s2 = mp.mpf(float(noise_var))
    Kuf = mp.matrix(M, N)
    Kuu = mp.matrix(M, M)
    kd = mp.mpf(0)
    for k in kern_list:
        Kuf += kern_K(k, zs, xs)
        Kuu += kern_K(k, zs, zs)
        kd += kern_Kdiag(k)
    for i in range(M):
        Kuu[i, i] += mp.mpf("1e-6")
    L = _chol(Kuu)
    A = _trsm_lower(L, Kuf) / mp.sqrt(s2)
    AAT = A * A.T
    B = AAT.copy()
    for i in range(M):
        B[i, i] += 1
    LB = _chol(B)
    err = mp.matrix([[v] for v in ys])
    c = _trsm_lower(LB, A * err) / mp.sqrt(s2)
    bound = -mp.mpf(N) / 2 * mp.log(2 * mp.pi)
    bound -= sum(mp.log(LB[i, i]) for i in range(M))
    bound -= mp.mpf(N) / 2 * mp.log(s2)
    bound -= sum(v ** 2 for v in ys) / (2 * s2)
    bound += sum(c[i, 0] ** 2 for i in range(M)) / 2
    bound -= N * kd / (2 * s2)
    bound += sum(AAT[i, i] for i in range(M)) / 2
    return float(bound)
