"""CPU restatement of the gpitch pdgp / sgpr_ss ELBO path (TEST INFRASTRUCTURE ONLY).

Parity pin: the reference's one printed deterministic ELBO result (demo_modgp-real-audio.ipynb) is reproduced through
these functions to 5.7e-7 relative after 10000 Adam steps (oracle/demo_anchor.py; see oracle/__init__.py for what
that run covers and what it does not).

Every function cites the reference lines it follows (paths relative to
/root/reference).  The arithmetic that lives in the reference's third-party
dependency **GPflow 0.5 / TensorFlow 1.2.1** (versions printed in
demos/notebooks/demo_modgp.ipynb:36-37; not vendored, not installable offline) is
restated from that release's published algorithm, anchored on the reference's call
sites: ``conditional`` (gpitch/pdgp.py:147-155), ``gauss_kl`` (gpitch/pdgp.py:120-129),
``Stationary.euclid_dist`` (gpitch/matern12_spectral_mixture.py:106),
``quadrature.hermgauss`` (gpitch/likelihoods.py:35), ``SGPR`` (gpitch/sgpr_ss.py:10,25).

All functions take an array backend ``xp`` (oracle/backend.py): numpy (default) for
golden values, torch-CPU for autograd gradients and the CPU baseline timing.
Kernels are plain dicts:  {"type": ..., "variance": v, "lengthscales": l,
"energy": [e_k], "frequency": [f_k]}.
"""
import numpy as np
from .backend import NP

JITTER = 1e-6            # gpflow settings.numerics.jitter_level (pdgp.py:14, sgpr_ss.py:43)
NUM_GH = 20              # likelihoods.py:435  (H = 20)

NLIN_LOGISTIC, NLIN_SOFTPLUS, NLIN_GAUSS = 0, 1, 2


# ----------------------------------------------------------------------------------
# nonlinearities — gpitch/methods.py:197-233
# ----------------------------------------------------------------------------------
def logistic(x, xp=NP):
    """methods.py:197-199 / 216-218: 1/(1+exp(-2(x-pi))) (shifted, scaled)."""
    return 1. / (1. + xp.exp(-2. * (x - np.pi)))


def softplus(x, xp=NP):
    """methods.py:205-207 / 220-222: log(exp(x)+1), the naive form."""
    return xp.log(xp.exp(x) + 1.)


def gaussfun(x, xp=NP):
    """methods.py:213-214 / 232-233: exp(-2(x-pi)^2)."""
    return xp.exp(-2. * (x - np.pi) ** 2)


def nlinfun(code):
    return {NLIN_LOGISTIC: logistic, NLIN_SOFTPLUS: softplus, NLIN_GAUSS: gaussfun}[code]


# ----------------------------------------------------------------------------------
# kernels
# ----------------------------------------------------------------------------------
def square_dist(X, X2, lengthscales, xp=NP):
    """GPflow 0.5 Stationary.square_dist (used via euclid_dist at m12sm.py:106):
    X/l; -2 X X2^T + |X|^2[:,None] + |X2|^2[None,:]  (matmul expansion, kept literal)."""
    X = X / lengthscales
    Xs = xp.sum(xp.square(X), 1)
    if X2 is None:
        return -2. * xp.matmul(X, xp.t(X)) + xp.reshape(Xs, (-1, 1)) + xp.reshape(Xs, (1, -1))
    X2 = X2 / lengthscales
    X2s = xp.sum(xp.square(X2), 1)
    return -2. * xp.matmul(X, xp.t(X2)) + xp.reshape(Xs, (-1, 1)) + xp.reshape(X2s, (1, -1))


def euclid_dist(X, X2, lengthscales, xp=NP):
    """GPflow 0.5 Stationary.euclid_dist: sqrt(r2 + 1e-12)  => r(x,x) = 1e-6."""
    return xp.sqrt(square_dist(X, X2, lengthscales, xp) + 1e-12)


def phi_features(kern, X, xp=NP):
    """matern12_spectral_mixture.py:123-133: [sqrt(e_k) cos(2 pi f_k x); sqrt(e_k) sin(...)] (2m x n).
    Features use the UNSCALED x (:129-130)."""
    n = X.shape[0]
    m = len(kern["frequency"])
    phi = 2 * m * [None]
    for i in range(m):
        e = xp.scalar(kern["energy"][i])
        f = xp.scalar(kern["frequency"][i])
        phi[i] = xp.sqrt(e) * xp.cos(2 * np.pi * f * X)
        phi[i + m] = xp.sqrt(e) * xp.sin(2 * np.pi * f * X)
    return xp.reshape(xp.stack(phi), (2 * m, n))


def K(kern, X, X2=None, xp=NP):
    """Kern.K(X, X2=None) for the kernel types on the path."""
    t = kern["type"]
    v = xp.scalar(kern["variance"])
    if t == "mercer_matern12sm":
        # matern12_spectral_mixture.py:102-117
        r = euclid_dist(X, X2, xp.scalar(kern["lengthscales"]), xp)
        phi = phi_features(kern, X, xp)
        phi2 = phi if X2 is None else phi_features(kern, X2, xp)
        return v * xp.exp(-r) * xp.matmul(xp.t(phi), phi2)
    if t == "matern12sm":
        # matern12_spectral_mixture.py:38-56 (broadcast form; r = sqrt((x-x'+1e-12)^2))
        if X2 is None:
            X2 = X
        r = xp.sqrt(xp.square(xp.reshape(X, (-1, 1)) - xp.reshape(X2, (1, -1)) + 1e-12))
        r1 = r / xp.scalar(kern["lengthscales"])
        k = xp.scalar(kern["energy"][0]) * xp.cos(2. * np.pi * xp.scalar(kern["frequency"][0]) * r)
        for i in range(1, len(kern["frequency"])):
            k = k + xp.scalar(kern["energy"][i]) * xp.cos(2. * np.pi * xp.scalar(kern["frequency"][i]) * r)
        return v * xp.exp(-r1) * k
    if t == "matern32sm":
        # gpitch/kernels.py:232-247: r = sqrt((x-x'+1e-12)^2); r1 = sqrt(3) r / l;
        # k = sum_i variance_i (1 + r1) exp(-r1) cos(2 pi f_i r)   (the dict's "energy" holds variance_i;
        # there is no global variance: kern["variance"] is 1)
        if X2 is None:
            X2 = X
        r = xp.sqrt(xp.square(xp.reshape(X, (-1, 1)) - xp.reshape(X2, (1, -1)) + 1e-12))
        r1 = np.sqrt(3.) * (r / xp.scalar(kern["lengthscales"]))
        k = xp.scalar(kern["energy"][0]) * (1. + r1) * xp.exp(-r1) * xp.cos(2. * np.pi * xp.scalar(kern["frequency"][0]) * r)
        for i in range(1, len(kern["frequency"])):
            k = k + xp.scalar(kern["energy"][i]) * (1. + r1) * xp.exp(-r1) * \
                xp.cos(2. * np.pi * xp.scalar(kern["frequency"][i]) * r)
        return v * k
    if t == "mercer_matern52sm":
        # Matern52 * MercerCosMix (init_models.py:183-198): GPflow Prod = elementwise product of
        # Matern52.K (variance v52, euclid_dist) and MercerCosMix.K = (phi * v_c)^T phi2 (kernels.py:358-370).
        # kern["variance"] holds v52 * v_c.
        r = euclid_dist(X, X2, xp.scalar(kern["lengthscales"]), xp)
        phi = phi_features(kern, X, xp)
        phi2 = phi if X2 is None else phi_features(kern, X2, xp)
        k52 = (1. + np.sqrt(5.) * r + 5. / 3. * xp.square(r)) * xp.exp(-np.sqrt(5.) * r)
        return v * k52 * xp.matmul(xp.t(phi), phi2)
    # GPflow 0.5 stationary kernels (call sites init_kernels.py:12, init_models.py:83,
    # demo-modgp.py:32)
    ls = xp.scalar(kern["lengthscales"])
    if t == "rbf":
        return v * xp.exp(-square_dist(X, X2, ls, xp) / 2.)
    r = euclid_dist(X, X2, ls, xp)
    if t == "matern12":
        return v * xp.exp(-r)
    if t == "matern32":
        return v * (1. + np.sqrt(3.) * r) * xp.exp(-np.sqrt(3.) * r)
    if t == "matern52":
        return v * (1. + np.sqrt(5.) * r + 5. / 3. * xp.square(r)) * xp.exp(-np.sqrt(5.) * r)
    raise ValueError("unknown kernel type %r" % (t,))


def Kdiag(kern, X, xp=NP):
    """Kern.Kdiag(X): exact fill, NOT diag(K(X)) (m12sm.py:58-62, :119-121; GPflow Stationary)."""
    n = X.shape[0]
    v = xp.scalar(kern["variance"])
    # (Matern32sm.Kdiag, kernels.py:249-253: sum of its variances; the Matern52 * MercerCosMix product has
    #  Kdiag = v52 * v_c: MercerCosMix.Kdiag fills its variance, kernels.py:372-373 — no energy sum)
    if kern["type"] in ("mercer_matern12sm", "matern12sm", "matern32sm"):
        s = xp.scalar(kern["energy"][0])
        for i in range(1, len(kern["energy"])):
            s = s + xp.scalar(kern["energy"][i])
        return xp.fill(n, v * s)
    return xp.fill(n, v)


def K_sum(kern_list, X, X2=None, xp=NP):
    """GPflow Add kernel: K = sum_p K_p (transcription.py:245, sgpr_ss.py:19-22)."""
    out = K(kern_list[0], X, X2, xp)
    for k in kern_list[1:]:
        out = out + K(k, X, X2, xp)
    return out


def Kdiag_sum(kern_list, X, xp=NP):
    out = Kdiag(kern_list[0], X, xp)
    for k in kern_list[1:]:
        out = out + Kdiag(k, X, xp)
    return out


# ----------------------------------------------------------------------------------
# GPflow 0.5 conditionals.conditional / kullback_leiblers.gauss_kl
# ----------------------------------------------------------------------------------
def conditional(Xnew, X, kern, f, q_sqrt=None, whiten=False, xp=NP, jitter=JITTER, full_cov=False):
    """conditional(Xnew, X, kern, f, full_cov, q_sqrt, whiten) with one latent column
    (call sites pdgp.py:147-155, 176-178, 185-187, 199-205 — all with full_cov=False).
    f: (M,1); q_sqrt: (M,M,1) full-matrix form, lower triangle forced by band_part.
    Returns fmean (N,1), fvar (N,1) — or (N,N,1) with full_cov=True (GPflow 0.5: fvar = K(Xnew) - A^T A + LTA^T LTA)."""
    M = X.shape[0]
    Kmn = K(kern, X, Xnew, xp)
    Kmm = K(kern, X, None, xp) + xp.eye(M) * jitter
    Lm = xp.cholesky(Kmm)
    A = xp.trsm(Lm, Kmn, lower=True)
    if full_cov:
        fvar = K(kern, Xnew, None, xp) - xp.matmul(xp.t(A), A)
    else:
        fvar = Kdiag(kern, Xnew, xp) - xp.sum(xp.square(A), 0)
    if not whiten:
        A = xp.trsm(xp.t(Lm), A, lower=False)
    fmean = xp.matmul(xp.t(A), f)
    if q_sqrt is not None:
        L = xp.tril(q_sqrt[:, :, 0])
        LTA = xp.matmul(xp.t(L), A)
        if full_cov:
            fvar = fvar + xp.matmul(xp.t(LTA), LTA)
        else:
            fvar = fvar + xp.sum(xp.square(LTA), 0)
    if full_cov:
        return fmean, xp.reshape(fvar, (fvar.shape[0], fvar.shape[1], 1))
    return fmean, xp.reshape(fvar, (-1, 1))


def gauss_kl(q_mu, q_sqrt, Kp=None, xp=NP):
    """gauss_kl(q_mu, q_sqrt, K=None) for one latent column (pdgp.py:120-121 whitened,
    :126-129 with K = Kuu + jitter I)."""
    M = q_mu.shape[0]
    Lq = xp.tril(q_sqrt[:, :, 0])
    if Kp is None:
        alpha = q_mu
        trace = xp.sum(xp.square(Lq))
    else:
        Lp = xp.cholesky(Kp)
        alpha = xp.trsm(Lp, q_mu, lower=True)
        trace = xp.sum(xp.square(xp.trsm(Lp, Lq, lower=True)))
    twoKL = xp.sum(xp.square(alpha)) - float(M) - xp.sum(xp.log(xp.square(xp.diag_part(Lq)))) + trace
    if Kp is not None:
        twoKL = twoKL + xp.sum(xp.log(xp.square(xp.diag_part(Lp))))
    return 0.5 * twoKL


# ----------------------------------------------------------------------------------
# modulated likelihood — gpitch/likelihoods.py:33-68, 422-447
# ----------------------------------------------------------------------------------
def hermgauss1d(mean_g, var_g, H, nlin, xp=NP):
    """likelihoods.py:33-45."""
    gh_x, gh_w = np.polynomial.hermite.hermgauss(H)
    gh_x = xp.asarray(gh_x.reshape(1, -1))
    gh_w = xp.asarray(gh_w.reshape(-1, 1) / np.sqrt(np.pi))
    X = gh_x * xp.sqrt(2. * var_g) + mean_g
    ev = nlin(X, xp)
    E1 = xp.reshape(xp.matmul(ev, gh_w), mean_g.shape)
    E2 = xp.reshape(xp.matmul(ev ** 2, gh_w), mean_g.shape)
    return E1, E2


def log_lik_exp(Y, mean_f, var_f, E1, E2, noise_var, Ksrc, xp=NP):
    """likelihoods.py:47-68."""
    A_l = [E1[i] * mean_f[i] for i in range(Ksrc)]
    B_l = [E2[i] * (var_f[i] + mean_f[i] ** 2) for i in range(Ksrc)]
    C_l = []
    for i in range(Ksrc - 1):
        for j in range(i + 1, Ksrc):
            C_l.append(E1[i] * mean_f[i] * E1[j] * mean_f[j])
    A = A_l[0]
    for a in A_l[1:]:
        A = A + a
    B = B_l[0]
    for b in B_l[1:]:
        B = B + b
    if Ksrc == 1:
        C = 0. * mean_f[0]
    else:
        C = C_l[0]
        for c in C_l[1:]:
            C = C + c
        C = 2. * C
    return -0.5 * ((1. / noise_var) * (Y ** 2 - 2. * Y * A + B + C) + np.log(2. * np.pi) + xp.log(noise_var))


def mpd_variational_expectations(Fmu, Fvar, Y, noise_var, num_sources, nlin_code=NLIN_LOGISTIC, xp=NP):
    """MpdLik.variational_expectations (likelihoods.py:422-447).  Column order of Fmu/Fvar:
    [g_0..g_{P-1}, f_0..f_{P-1}] (pdgp.py:157-164)."""
    nlin = nlinfun(nlin_code)
    P = num_sources
    mean_g = [xp.reshape(Fmu[:, i], (-1, 1)) for i in range(P)]
    mean_f = [xp.reshape(Fmu[:, i + P], (-1, 1)) for i in range(P)]
    var_g = [xp.reshape(Fvar[:, i], (-1, 1)) for i in range(P)]
    var_f = [xp.reshape(Fvar[:, i + P], (-1, 1)) for i in range(P)]
    E1, E2 = P * [None], P * [None]
    for i in range(P):
        E1[i], E2[i] = hermgauss1d(mean_g[i], var_g[i], NUM_GH, nlin, xp)
    return log_lik_exp(Y, mean_f, var_f, E1, E2, noise_var, P, xp)


# ----------------------------------------------------------------------------------
# Pdgp — gpitch/pdgp.py:113-208
# ----------------------------------------------------------------------------------
def pdgp_conditionals(x, za, zc, kern_act, kern_com, q_mu_act, q_sqrt_act, q_mu_com, q_sqrt_com,
                      whiten=True, xp=NP):
    """pdgp.py:146-164: 2P conditionals, concatenated [act..., com...]."""
    P = len(kern_act)
    ma, va, mc, vc = [], [], [], []
    for i in range(P):
        m, v = conditional(x, za[i], kern_act[i], q_mu_act[i], q_sqrt_act[i], whiten, xp)
        ma.append(m); va.append(v)
        m, v = conditional(x, zc[i], kern_com[i], q_mu_com[i], q_sqrt_com[i], whiten, xp)
        mc.append(m); vc.append(v)
    fmean = xp.concat([xp.concat(ma, 1), xp.concat(mc, 1)], 1)
    fvar = xp.concat([xp.concat(va, 1), xp.concat(vc, 1)], 1)
    return fmean, fvar


def pdgp_prior_kl(za, zc, kern_act, kern_com, q_mu_act, q_sqrt_act, q_mu_com, q_sqrt_com,
                  whiten=True, xp=NP):
    """Pdgp.build_prior_kl (pdgp.py:113-131)."""
    P = len(kern_act)
    kl = 0.
    for i in range(P):
        if whiten:
            kl = kl + gauss_kl(q_mu_act[i], q_sqrt_act[i], None, xp)
            kl = kl + gauss_kl(q_mu_com[i], q_sqrt_com[i], None, xp)
        else:
            ka = K(kern_act[i], za[i], None, xp) + xp.eye(za[i].shape[0]) * JITTER
            kc = K(kern_com[i], zc[i], None, xp) + xp.eye(zc[i].shape[0]) * JITTER
            kl = kl + gauss_kl(q_mu_act[i], q_sqrt_act[i], ka, xp)
            kl = kl + gauss_kl(q_mu_com[i], q_sqrt_com[i], kc, xp)
    return kl


def pdgp_elbo(x, y, za, zc, kern_act, kern_com, q_mu_act, q_sqrt_act, q_mu_com, q_sqrt_com,
              noise_var, num_data=None, whiten=True, nlin_code=NLIN_LOGISTIC, xp=NP):
    """Pdgp.build_likelihood (pdgp.py:133-170) on the batch (x, y)."""
    P = len(kern_act)
    kl = pdgp_prior_kl(za, zc, kern_act, kern_com, q_mu_act, q_sqrt_act, q_mu_com, q_sqrt_com, whiten, xp)
    fmean, fvar = pdgp_conditionals(x, za, zc, kern_act, kern_com, q_mu_act, q_sqrt_act,
                                    q_mu_com, q_sqrt_com, whiten, xp)
    var_exp = mpd_variational_expectations(fmean, fvar, y, noise_var, P, nlin_code, xp)
    nb = x.shape[0]
    scale = float(nb if num_data is None else num_data) / float(nb)
    return xp.sum(var_exp) * scale - kl


def pdgp_predict_act_n_com(xnew, za, zc, kern_act, kern_com, q_mu_act, q_sqrt_act, q_mu_com, q_sqrt_com,
                           whiten=True, nlin_code=NLIN_LOGISTIC, xp=NP):
    """Pdgp.predict_act_n_com (pdgp.py:190-208)."""
    nlin = nlinfun(nlin_code)
    P = len(kern_act)
    mean_a, var_a, mean_c, var_c, mean_s = [], [], [], [], []
    for i in range(P):
        m, v = conditional(xnew, za[i], kern_act[i], q_mu_act[i], q_sqrt_act[i], whiten, xp)
        mean_a.append(m); var_a.append(v)
        m, v = conditional(xnew, zc[i], kern_com[i], q_mu_com[i], q_sqrt_com[i], whiten, xp)
        mean_c.append(m); var_c.append(v)
        mean_s.append(nlin(mean_a[i], xp) * mean_c[i])
    return mean_a, var_a, mean_c, var_c, mean_s


# ----------------------------------------------------------------------------------
# SGPRSS — gpitch/sgpr_ss.py:29-106 (+ GPflow 0.5 SGPR.build_predict)
# ----------------------------------------------------------------------------------
def sgpr_common(X, Y, Z, kern_list, noise_var, xp=NP):
    """sgpr_ss.py:40-53 intermediates (zero mean function)."""
    M = Z.shape[0]
    err = Y
    Kdg = Kdiag_sum(kern_list, X, xp)
    Kuf = K_sum(kern_list, Z, X, xp)
    Kuu = K_sum(kern_list, Z, None, xp) + xp.eye(M) * JITTER
    L = xp.cholesky(Kuu)
    sigma = xp.sqrt(noise_var)
    A = xp.trsm(L, Kuf, lower=True) / sigma
    AAT = xp.matmul(A, xp.t(A))
    B = AAT + xp.eye(M)
    LB = xp.cholesky(B)
    Aerr = xp.matmul(A, err)
    c = xp.trsm(LB, Aerr, lower=True) / sigma
    return err, Kdg, L, A, AAT, LB, c


def sgpr_bound(X, Y, Z, kern_list, noise_var, reg=False, xp=NP):
    """SGPRSS.build_likelihood (sgpr_ss.py:29-71)."""
    num_data = float(Y.shape[0])
    output_dim = float(Y.shape[1])
    err, Kdg, L, A, AAT, LB, c = sgpr_common(X, Y, Z, kern_list, noise_var, xp)
    bound = -0.5 * num_data * output_dim * np.log(2 * np.pi)
    bound = bound - output_dim * xp.sum(xp.log(xp.diag_part(LB)))
    bound = bound - 0.5 * num_data * output_dim * xp.log(noise_var)
    bound = bound - 0.5 * xp.sum(xp.square(err)) / noise_var
    bound = bound + 0.5 * xp.sum(xp.square(c))
    bound = bound - 0.5 * output_dim * xp.sum(Kdg) / noise_var
    bound = bound + 0.5 * output_dim * xp.sum(xp.diag_part(AAT))
    if reg:
        beta = 1000.
        s = xp.abs(xp.scalar(kern_list[0]["variance"]))
        for k in kern_list[1:]:
            s = s + xp.abs(xp.scalar(k["variance"]))
        bound = bound - beta * s
    return bound


def sgpr_predict_f(Xnew, X, Y, Z, kern_list, noise_var, xp=NP, full_cov=False):
    """GPflow 0.5 SGPR.build_predict (predict_f, called separation.py:306 with full_cov=False; full_cov=True:
    var = K(Xnew) + tmp2^T tmp2 - tmp1^T tmp1, tiled to n x n x D)."""
    err, Kdg, L, A, AAT, LB, c = sgpr_common(X, Y, Z, kern_list, noise_var, xp)
    Kus = K_sum(kern_list, Z, Xnew, xp)
    tmp1 = xp.trsm(L, Kus, lower=True)
    tmp2 = xp.trsm(LB, tmp1, lower=True)
    mean = xp.matmul(xp.t(tmp2), c)
    D = Y.shape[1]
    if full_cov:
        var = K_sum(kern_list, Xnew, None, xp) + xp.matmul(xp.t(tmp2), tmp2) - xp.matmul(xp.t(tmp1), tmp1)
        return mean, _tile_last(xp.reshape(var, (var.shape[0], var.shape[1], 1)), D, xp)
    var = Kdiag_sum(kern_list, Xnew, xp) + xp.sum(xp.square(tmp2), 0) - xp.sum(xp.square(tmp1), 0)
    return mean, _tile_last(xp.reshape(var, (-1, 1)), D, xp)


def _tile_last(a, D, xp=NP):
    """tf.tile over the trailing output axis (GPflow 0.5 SGPR.build_predict; sgpr_ss.py:97-98,102)"""
    return a if D == 1 else xp.concat([a] * D, axis=a.ndim - 1)


def sgpr_predict_source(Xnew, X, Y, kern_list, noise_var, xp=NP, full_cov=False):
    """SGPRSS.build_predict_source (sgpr_ss.py:73-106): exact GP, N x N Cholesky; the variance uses
    the SUM kernel's Kdiag (:101) — reproduced as is."""
    N = X.shape[0]
    Kxx = K_sum(kern_list, X, None, xp) + xp.eye(N) * noise_var
    L = xp.cholesky(Kxx)
    V = xp.trsm(L, Y, lower=True)
    means, variances = [], []
    D = Y.shape[1]
    for kp in kern_list:
        Kx = K(kp, X, Xnew, xp)
        A = xp.trsm(L, Kx, lower=True)
        means.append(xp.matmul(xp.t(A), V))
        if full_cov:     # sgpr_ss.py:95-99: the SUM kernel's K(Xnew), as the diagonal form uses its Kdiag
            svar = K_sum(kern_list, Xnew, None, xp) - xp.matmul(xp.t(A), A)
            variances.append(_tile_last(xp.reshape(svar, (svar.shape[0], svar.shape[1], 1)), D, xp))
            continue
        svar = Kdiag_sum(kern_list, Xnew, xp) - xp.sum(xp.square(A), 0)
        variances.append(_tile_last(xp.reshape(svar, (-1, 1)), D, xp))
    return means, variances


# ----------------------------------------------------------------------------------
# GPflow 0.5 runtime pieces on the path
# ----------------------------------------------------------------------------------
def positive_forward(x):
    """transforms.positive = Log1pe(lower=1e-6): y = log(1+e^x) + 1e-6 (numerically stable form)."""
    x = np.asarray(x, dtype=np.float64)
    return np.logaddexp(0., x) + 1e-6


def positive_backward(y):
    y = np.asarray(y, dtype=np.float64) - 1e-6
    return y + np.log(-np.expm1(-y))


def minibatch_indices(rng, N, mb):
    """gpflow.minibatch.MinibatchData index generation: with replacement when mb/N < 0.5,
    otherwise a permutation prefix (seeded RandomState(0) at pdgp.py:76-77)."""
    if float(mb) / float(N) < 0.5:
        return rng.randint(N, size=mb)
    return rng.permutation(N)[:mb]


def adam_step(x, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update on the free-state vector (TF 1.2 'epsilon hat' form):
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMA; x -= lr_t*m/(sqrt(v)+eps).  g is d(-ELBO)/dx."""
    m = beta1 * m + (1. - beta1) * g
    v = beta2 * v + (1. - beta2) * g * g
    lr_t = lr * np.sqrt(1. - beta2 ** t) / (1. - beta1 ** t)
    x = x - lr_t * m / (np.sqrt(v) + eps)
    return x, m, v
