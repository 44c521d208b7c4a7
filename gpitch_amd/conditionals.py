"""GPflow-0.5 operators the reference calls by name: conditionals.conditional
(gpitch/pdgp.py:147-155,176-205) and kullback_leiblers.gauss_kl (gpitch/pdgp.py:120-129).
Host arrays in, host arrays out (as AutoFlow does); arithmetic in csrc/ via the C-ABI."""
import ctypes as C

import numpy as np

from . import _lib


def _kdesc(h, kern):
    th = h.to_device(kern.theta())
    return _lib.KernelDesc(kern.type_code, int(getattr(kern, "num_partials", 0) or 0), th.data_ptr()), th


def conditional(Xnew, X, kern, f, full_cov=False, q_sqrt=None, whiten=False, float_type=None):
    """mean and variance of f(Xnew) given q(u) = N(f, q_sqrt q_sqrt^T) at inducing inputs X.
    full_cov=True: the variance is the N x N x 1 posterior covariance (GPflow's shape for one latent column).
    float_type=np.float32: the strips and strip products in float32 (gp_conditional_diag_f32)."""
    h = _lib.default_handle()
    Xnew = np.asarray(Xnew, dtype=np.float64).reshape(-1, 1)
    X = np.asarray(X, dtype=np.float64).reshape(-1, 1)
    f = np.asarray(f, dtype=np.float64)
    if f.ndim != 2 or f.shape[1] != 1:
        raise ValueError("one latent column per call (f must be M x 1)")
    N, M = Xnew.shape[0], X.shape[0]
    d, th = _kdesc(h, kern)
    dsq = None
    if q_sqrt is not None:
        q = np.asarray(q_sqrt, dtype=np.float64)
        if q.ndim == 3:
            q = q[:, :, 0]
        elif q.ndim == 2 and q.shape[1] == 1 and M != 1:
            q = np.diag(q[:, 0])          # GPflow's diagonal (M x K) q_sqrt form
        dsq = h.to_device(q)
    dx, dz, dmu = h.to_device(Xnew), h.to_device(X), h.to_device(f)
    if full_cov:
        if _lib.precision_bits(float_type) == 32:
            raise ValueError("full_cov=True is a float64 operator")
        fm, fc = h.empty(max(N, 1)), h.empty(max(N, 1), max(N, 1))
        ws = h.workspace(h.lib.gp_conditional_full_workspace_bytes(N, M))
        h.check(h.lib.gp_conditional_full(h.h, C.byref(d), dx.data_ptr(), N, dz.data_ptr(), M, dmu.data_ptr(),
                                          None if dsq is None else dsq.data_ptr(), int(bool(whiten)), 1e-6,
                                          fm.data_ptr(), fc.data_ptr(), ws.data_ptr(), ws.numel()))
        return fm[:N].cpu().numpy().reshape(-1, 1), fc[:N, :N].cpu().numpy().reshape(N, N, 1)
    fm, fv = h.empty(max(N, 1)), h.empty(max(N, 1))
    ws = h.workspace(h.lib.gp_conditional_workspace_bytes(N, M))
    if _lib.precision_bits(float_type) == 32:
        h.check(h.lib.gp_conditional_diag_f32w(h.h, C.byref(d), dx.data_ptr(), N, dz.data_ptr(), M, dmu.data_ptr(),
                                               None if dsq is None else dsq.data_ptr(), int(bool(whiten)), 1e-6,
                                               fm.data_ptr(), fv.data_ptr(), ws.data_ptr(), ws.numel()))
        return fm[:N].cpu().numpy().reshape(-1, 1), fv[:N].cpu().numpy().reshape(-1, 1)
    h.check(h.lib.gp_conditional_diag(h.h, C.byref(d), dx.data_ptr(), N, dz.data_ptr(), M, dmu.data_ptr(),
                                      None if dsq is None else dsq.data_ptr(), int(bool(whiten)), 1e-6,
                                      fm.data_ptr(), fv.data_ptr(), ws.data_ptr(), ws.numel()))
    return fm[:N].cpu().numpy().reshape(-1, 1), fv[:N].cpu().numpy().reshape(-1, 1)


def gauss_kl(q_mu, q_sqrt, K=None):
    """KL[q(u) || p(u)]: whitened (p = N(0, I)) when K is None (pdgp.py:120-121); p = N(0, K) for an explicit
    M x M matrix K (pdgp.py:126-129 passes kern.K(z) + jitter I)."""
    h = _lib.default_handle()
    q_mu = np.asarray(q_mu, dtype=np.float64)
    q = np.asarray(q_sqrt, dtype=np.float64)
    if q.ndim == 3:
        q = q[:, :, 0]
    M = q_mu.shape[0]
    out = C.c_double()
    ws = h.workspace(h.lib.gp_gauss_kl_workspace_bytes(M, int(K is not None)))
    dmu, dq = h.to_device(q_mu), h.to_device(q)   # named so they outlive the call
    if K is None:
        h.check(h.lib.gp_gauss_kl(h.h, dmu.data_ptr(), dq.data_ptr(), M, None, None, 1e-6,
                                  C.byref(out), ws.data_ptr(), ws.numel()))
    else:
        dK = h.to_device(np.ascontiguousarray(np.asarray(K, dtype=np.float64).reshape(M, M)))
        h.check(h.lib.gp_gauss_kl_matrix(h.h, dmu.data_ptr(), dq.data_ptr(), M, dK.data_ptr(),
                                         C.byref(out), ws.data_ptr(), ws.numel()))
    return out.value
