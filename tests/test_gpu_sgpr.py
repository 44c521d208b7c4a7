"""GPU parity of the SGPRSS path (bound, predict_f, predict_s) against the oracle and the golden vector."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gpflow05 as orc  # noqa: E402


def _model(X, Y, Z, kdicts, noise, handle, reg=False, float_type=None):
    import gpitch_amd
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.sgpr_ss import SGPRSS
    ks = [MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=d["variance"],
                           lengthscales=d["lengthscales"]) for d in kdicts]
    m = SGPRSS(X, Y, np.sum(ks), Z, reg=reg, handle=handle, float_type=float_type)
    m.likelihood.variance = noise
    return m


def _problem(N, M, P, seed, npart=2):
    rng = np.random.RandomState(seed)
    fs = 16000.
    X = np.linspace(0, (N - 1) / fs, N).reshape(-1, 1)
    kl = []
    Y = np.zeros((N, 1))
    for p in range(P):
        f0 = 220. * 2 ** (p * 4 / 12.)
        Y += np.sin(2 * np.pi * f0 * X) * np.exp(-((X - X.mean()) / (0.3 * np.ptp(X) + 1e-9)) ** 2)
        kl.append({"type": "mercer_matern12sm", "variance": 1.0 + 0.1 * p, "lengthscales": 0.05 + 0.02 * p,
                   "energy": [0.6, 0.4] if npart == 2 else list(np.linspace(1.0, 0.2, npart) / npart),
                   "frequency": [f0 * (q + 1) for q in range(npart)]})
    Y += 0.05 * rng.randn(N, 1)
    Z = X[:: max(N // M, 1)][:M].copy()
    return X, Y, Z, kl


@pytest.mark.parametrize("N,M,P", [(200, 12, 1), (2001, 64, 3), (3000, 130, 5)])
@pytest.mark.parametrize("reg", [False, True])
def test_sgpr_bound_matches_oracle(gp_handle, N, M, P, reg):
    X, Y, Z, kl = _problem(N, M, P, N)
    m = _model(X, Y, Z, kl, 0.3, gp_handle, reg=reg)
    got = m.build_likelihood()
    ref = orc.sgpr_bound(X, Y, Z, kl, 0.3, reg=reg)
    assert abs(got - ref) <= 1e-9 * abs(ref), (got, ref)


def test_sgpr_bound_matches_golden(gp_handle):
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sgpr_small.npz"))
    kl = []
    for i in range(int(d["P"])):
        th = d["kern_%d_theta" % i]
        mm = (len(th) - 2) // 2
        kl.append({"type": "mercer_matern12sm", "variance": th[0], "lengthscales": th[1], "energy": list(th[2:2 + mm]),
                   "frequency": list(th[2 + mm:])})
    m = _model(d["X"], d["Y"], d["Z"], kl, float(d["noise_var"]), gp_handle)
    got = m.build_likelihood()
    assert abs(got - float(d["bound"])) <= 1e-9 * abs(float(d["bound"]))


def test_sgpr_predictions_match_oracle(gp_handle):
    X, Y, Z, kl = _problem(1500, 50, 3, 7)
    m = _model(X, Y, Z, kl, 0.2, gp_handle)
    Xs = X[::7] + 1e-5
    mean, var = m.predict_f(Xs)
    rm, rv = orc.sgpr_predict_f(Xs, X, Y, Z, kl, 0.2)
    np.testing.assert_allclose(mean, rm, rtol=0, atol=1e-8 * np.abs(rm).max())
    np.testing.assert_allclose(var, rv, rtol=0, atol=1e-8 * np.abs(rv).max())
    sm, sv = m.predict_s(Xs)
    rsm, rsv = orc.sgpr_predict_source(Xs, X, Y, kl, 0.2)
    for i in range(3):
        np.testing.assert_allclose(sm[i], rsm[i], rtol=0, atol=1e-8 * max(np.abs(rsm[i]).max(), 1e-3))
        np.testing.assert_allclose(sv[i], rsv[i], rtol=0, atol=1e-8 * np.abs(rsv[i]).max())


def test_sgpr_window_swap(gp_handle):
    """AMT/SoSp reset the DataHolders per window (transcription.py:253-263); the plan must follow."""
    X, Y, Z, kl = _problem(400, 20, 2, 1)
    m = _model(X, Y, Z, kl, 0.5, gp_handle)
    b0 = m.build_likelihood()
    X2, Y2, Z2, _ = _problem(500, 25, 2, 2)
    m.X = X2
    m.Y = 20. * Y2
    m.Z = Z2
    m.likelihood.variance = 1.
    b1 = m.build_likelihood()
    ref = orc.sgpr_bound(X2, 20. * Y2, Z2, kl, 1.0)
    assert abs(b1 - ref) <= 1e-9 * abs(ref) and b0 != b1


def _torch_bound_and_grads(X, Y, Z, kl, noise, reg=False):
    import torch
    from oracle.backend import TorchBackend
    tb = TorchBackend()
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)
    nv = T(noise)
    leaves = [nv]
    tk = []
    for d in kl:
        o = dict(d)
        o["variance"] = T(d["variance"]); o["lengthscales"] = T(d["lengthscales"])
        o["energy"] = [T(e) for e in d["energy"]]; o["frequency"] = [T(f) for f in d["frequency"]]
        leaves += [o["variance"], o["lengthscales"]] + o["energy"] + o["frequency"]
        tk.append(o)
    b = orc.sgpr_bound(torch.tensor(X), torch.tensor(Y), torch.tensor(Z), tk, nv, reg=reg, xp=tb)
    b.backward()
    return float(b.detach()), np.array([float(l.grad) for l in leaves])


@pytest.mark.parametrize("N,M,P,reg,npart", [(300, 16, 1, False, 2), (1200, 48, 3, True, 2), (1100, 40, 5, False, 4),
                                             (900, 32, 3, False, 6), (1500, 96, 2, False, 3), (4500, 160, 2, True, 2)])
def test_sgpr_gradient_matches_autograd(gp_handle, N, M, P, reg, npart):
    """P >= 2 Mercer kernels: Kuu / Kuf of the sum come from one pass (cov_mercer_sum_kernel: padded partial counts 4 and
    8) and, with at most four partials per kernel, the Kuf-side contractions of all kernels from one pass over Kuf_bar
    (hyper_contract_sum_kernel); six partials take the per-kernel contraction; M = 96 / 160: the M x M chain's products on 32 x 32
    tiles (gemm_tile32), M = 160 with N >= 4096 also the cluster factorisation of Kuu and B"""
    X, Y, Z, kl = _problem(N, M, P, N + 1, npart)
    m = _model(X, Y, Z, kl, 0.3, gp_handle, reg=reg)
    m._compile(); m._pack()
    ps = m._param_list()
    x0 = np.array([p.transform.backward(p.value)[0] for p in ps])
    f, gfree = m._objective(x0)
    ref_b, ref_g = _torch_bound_and_grads(X, Y, Z, kl, 0.3, reg=reg)
    assert abs(-f - ref_b) <= 1e-9 * abs(ref_b)
    got = -gfree * (1. + np.exp(-x0))        # undo the positive-transform chain rule: d/d constrained
    np.testing.assert_allclose(got, ref_g, rtol=0, atol=2e-7 * np.abs(ref_g).max())


def test_sgpr_optimize_increases_bound(gp_handle):
    X, Y, Z, kl = _problem(800, 32, 2, 11)
    m = _model(X, Y, Z, kl, 1.0, gp_handle)
    for k in m.kern.kern_list:            # as the reference's component kernels: lengthscale fixed, rest free
        k.lengthscales.fixed = True
    b0 = m.build_likelihood()
    res = m.optimize(maxiter=15)
    b1 = m.build_likelihood()
    assert b1 > b0 and np.isfinite(res.fun) and abs(-res.fun - b1) <= 1e-8 * abs(b1)
    assert m.kern.kern_list[0].lengthscales.value[0] == kl[0]["lengthscales"]


def test_fit_windows_streams_match_sequential_loop(gp_handle):
    """gpitch_amd.windows.fit_windows (the AMT / SoSp window loop, transcription.py:265-288) on several HIP
    streams: every window's result equals a plain sequential loop over the same windows, whatever the number of
    streams, and overlap-added per-window predictions can be produced from the worker."""
    from gpitch_amd.windows import fit_windows, default_reset
    nwin, N, M, P = 6, 400, 20, 2
    probs = [_problem(N, M, P, seed=20 + w) for w in range(nwin)]
    X0, Y0, Z0, kl0 = probs[0]
    make = lambda h: _model(X0, Y0, Z0, kl0, 1.0, h)
    wins = [(p[0], 20. * p[1], p[2]) for p in probs]          # transcription.py:255 scales y by 20

    def after(model, idx):
        mean, var = model.predict_f(wins[idx][0][::50])
        return {"bound": model.build_likelihood(), "var": np.array([k.variance.value[0] for k in model.kern.kern_list]),
                "noise": model.likelihood.variance.value[0], "mean": mean}

    # sequential reference loop on the test's own handle
    seq = []
    for idx, (x, y, z) in enumerate(wins):
        m = make(gp_handle)
        default_reset(m, x, y, z)
        m.optimize(maxiter=8)
        seq.append(after(m, idx))
    for ns in (1, 3):
        got = fit_windows(make, wins, maxiter=8, num_streams=ns, after_fit=after)
        assert len(got) == nwin
        for a, b in zip(got, seq):
            assert abs(a["bound"] - b["bound"]) <= 1e-9 * abs(b["bound"])
            np.testing.assert_allclose(a["var"], b["var"], rtol=1e-8)
            np.testing.assert_allclose(a["mean"], b["mean"], rtol=1e-7, atol=1e-9)
    # windows dealt over two ranks: each rank returns its own, None elsewhere
    r0 = fit_windows(make, wins, maxiter=2, num_streams=2, rank=0, world_size=2)
    r1 = fit_windows(make, wins, maxiter=2, num_streams=2, rank=1, world_size=2)
    assert [r is not None for r in r0] == [True, False] * 3 and [r is not None for r in r1] == [False, True] * 3
    assert r0[0]["bound"] > -1e300 and r0[0]["nfev"] >= 1


def test_sgpr_graph_replay_is_bitwise_eager():
    """gp_sgpr_bound_grad records its launch sequence into a hipGraph from the second call with the same buffers
    (handle on its own stream): replayed evaluations must reproduce the eager ones bit for bit, also after the
    parameters change in place and after a prediction call invalidated the recording."""
    import ctypes as C
    import torch
    from gpitch_amd import _lib
    X, Y, Z, kl = _problem(700, 24, 2, seed=31)
    eager_model = _model(X, Y, Z, kl, 0.7, _lib.default_handle())     # null stream: never captured
    eager_model._compile(); eager_model._pack()
    ps = eager_model._param_list()
    x0 = np.array([p.transform.backward(p.value)[0] for p in ps if not p.fixed])
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        h = _lib.Handle(torch.cuda.current_device(), stream=s)
        m = _model(X, Y, Z, kl, 0.7, h)
        m._compile(); m._pack()
        for step in range(5):
            xs = x0 + 0.01 * step
            f_e, g_e = eager_model._objective(xs)
            f_g, g_g = m._objective(xs)
            assert f_e == f_g and np.array_equal(g_e, g_g), step
        m.predict_f(X[::10])                      # rewrites a descriptor block -> recording dropped, re-captured
        for step in range(3):
            xs = x0 - 0.02 * step
            f_e, g_e = eager_model._objective(xs)
            f_g, g_g = m._objective(xs)
            assert f_e == f_g and np.array_equal(g_e, g_g), step
        c = [C.c_int64() for _ in range(3)]
        h.check(h.lib.gp_sgpr_eval_counts(m._plan, *[C.byref(v) for v in c]))
        assert c[1].value == 2 and c[2].value == 4 and c[0].value == 2, [v.value for v in c]
        m._destroy()
        s.synchronize()
        h.close()


@pytest.mark.parametrize("world,reg", [(2, False), (3, True)])
def test_frame_sharded_window_matches_unsharded(gp_handle, world, reg):
    """ONE window sharded over its frames (gp_sgpr_bound_begin -> sum of the M^2+M+2 exchange vectors ->
    gp_sgpr_bound_end -> sum of the gradient vectors), the ranks emulated in this process: every rank must report
    the unsharded bound, and the summed gradient must equal the unsharded gradient."""
    import ctypes as C
    X, Y, Z, kl = _problem(1101, 40, 3, seed=41)
    h = gp_handle
    full = _model(X, Y, Z, kl, 0.6, h, reg=reg)
    full._compile(); full._pack()
    g_full = h.zeros(full._nparams)
    b_full = full._bound(g_full)
    from gpitch_amd.sgpr_ss import SGPRSS
    shards = []
    for r in range(world):
        m = _model(X, Y, Z, kl, 0.6, h, reg=reg)
        object.__setattr__(m, "_shard", (r, world))
        m._compile(); m._pack()
        shards.append(m)
    assert sum(m._n_local for m in shards) == 1101
    need = int(h.lib.gp_sgpr_exchange_doubles(shards[0]._plan))
    assert need == 40 * 40 + 40 + 2
    parts = []
    for m in shards:
        x = h.empty(need)
        h.check(h.lib.gp_sgpr_bound_begin(m._plan, m._params.data_ptr(), m._Xd.data_ptr(), m._Yd.data_ptr(), m._n_local,
                                          m._Zd.data_ptr(), x.data_ptr()))
        parts.append(x)
    total = sum(parts)
    gsum = h.zeros(full._nparams)
    for r, m in enumerate(shards):
        out = C.c_double()
        g = h.zeros(full._nparams)
        h.check(h.lib.gp_sgpr_bound_end(m._plan, m._params.data_ptr(), m._Xd.data_ptr(), m._Yd.data_ptr(), m._n_local, 1101,
                                        m._Zd.data_ptr(), total.data_ptr(), m._bound_dev.data_ptr(), C.byref(out),
                                        g.data_ptr(), int(r == 0)))
        assert abs(out.value - b_full) <= 1e-10 * abs(b_full), (r, out.value, b_full)
        gsum += g
    gf, gs = g_full.cpu().numpy(), gsum.cpu().numpy()
    np.testing.assert_allclose(gs, gf, rtol=1e-8, atol=1e-9 * np.abs(gf).max())
    # d bound / d err (gp_sgpr_residual_grad: what trainable mean-function Params are differentiated through): each rank's
    # slice of it is the unsharded vector's slice
    r_full = h.empty(1101)
    h.check(h.lib.gp_sgpr_residual_grad(full._plan, full._params.data_ptr(), full._Yd.data_ptr(), 1101, r_full.data_ptr()))
    pieces = []
    for m in shards:
        rr = h.empty(m._n_local)
        h.check(h.lib.gp_sgpr_residual_grad(m._plan, m._params.data_ptr(), m._Yd.data_ptr(), m._n_local, rr.data_ptr()))
        pieces.append(rr.cpu().numpy())
    rf = r_full.cpu().numpy()
    np.testing.assert_allclose(np.concatenate(pieces), rf, rtol=0, atol=1e-9 * np.abs(rf).max())
    # end without a matching begin is refused
    with pytest.raises(Exception):
        m = shards[0]
        h.check(h.lib.gp_sgpr_bound_end(m._plan, m._params.data_ptr(), m._Xd.data_ptr(), m._Yd.data_ptr(), m._n_local, 1101,
                                        m._Zd.data_ptr(), total.data_ptr(), m._bound_dev.data_ptr(), None, None, 1))


def test_sgprss_mean_function(gp_handle):
    """SGPRSS(mean_function=...) (sgpr_ss.py:14,40,90,95): bound and gradients are those of the data with the mean
    subtracted; predicted means get it back (each source's too, as the reference does)."""
    from gpitch_amd.mean_functions import Constant, Linear, Zero
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.sgpr_ss import SGPRSS
    X, Y, Z, kl = _problem(1200, 40, 2, 11)
    for mf in (Zero(), Constant(0.37), Linear(2.0, -0.1), lambda x: 0.2 * np.sin(40. * x)):
        ks = [MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=d["variance"],
                               lengthscales=d["lengthscales"]) for d in kl]
        m = SGPRSS(X, Y, np.sum(ks), Z, mean_function=mf, handle=gp_handle)
        m.likelihood.variance = 0.3
        shift = np.asarray(mf(X)).reshape(-1, 1)
        ref = orc.sgpr_bound(X, Y - shift, Z, kl, 0.3)
        got = m.build_likelihood()
        assert abs(got - ref) <= 1e-9 * abs(ref), (got, ref)
        plain = _model(X, Y - shift, Z, kl, 0.3, gp_handle)
        xs = X[::7]
        mu, var = m.predict_f(xs)
        mu0, var0 = plain.predict_f(xs)
        np.testing.assert_allclose(mu, mu0 + np.asarray(mf(xs)).reshape(-1, 1), rtol=0, atol=1e-12)
        np.testing.assert_allclose(var, var0, rtol=0, atol=1e-12)
        ms, vs = m.predict_s(xs)
        ms0, vs0 = plain.predict_s(xs)
        for a, b in zip(ms, ms0):
            np.testing.assert_allclose(a, b + np.asarray(mf(xs)).reshape(-1, 1), rtol=0, atol=1e-11)
    with pytest.raises(TypeError):
        SGPRSS(X, Y, np.sum(ks), Z, mean_function=3.0, handle=gp_handle)


@pytest.mark.parametrize("ft", [np.float64, np.float32])
def test_sgprss_trainable_mean_function(gp_handle, ft):
    """GPflow mean functions carry trainable Params (Constant.c, Linear.A / .b; sgpr_ss.py:14,25,40): their gradient comes from
    d bound / d err on the device (gp_sgpr_residual_grad) and the chain rule on the host.  Residual gradient and the A, b
    entries against torch autograd through the oracle; then L-BFGS-B moves them (a data offset the zero-mean GP cannot
    explain cheaply) and improves the bound beyond what the kernel parameters alone reach."""
    import torch
    from oracle.backend import TorchBackend
    from gpitch_amd.mean_functions import Linear, Constant
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.sgpr_ss import SGPRSS
    X, Y, Z, kl = _problem(1200, 40, 2, 11)
    Y = Y + 0.8 + 1.5 * X                                  # an offset and a trend in the data
    mk = lambda: [MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=d["variance"],
                                   lengthscales=d["lengthscales"]) for d in kl]
    mf = Linear(0.4, 0.1)
    m = SGPRSS(X, Y, np.sum(mk()), Z, mean_function=mf, handle=gp_handle, float_type=ft)
    m.likelihood.variance = 0.3
    m._compile(); m._pack()
    ps = m._param_list()
    assert ps[-2] is mf.A and ps[-1] is mf.b
    x0 = np.array([p.transform.backward(p.value)[0] for p in ps])
    f, gfree = m._objective(x0)
    # reference: autograd through the oracle with A, b (and Y) as leaves
    tb = TorchBackend()
    A_t = torch.tensor(0.4, dtype=torch.float64, requires_grad=True)
    b_t = torch.tensor(0.1, dtype=torch.float64, requires_grad=True)
    Y_t = torch.tensor(Y, requires_grad=True)
    X_t = torch.tensor(X)
    bound = orc.sgpr_bound(X_t, Y_t - (A_t * X_t + b_t), torch.tensor(Z), kl, torch.tensor(0.3, dtype=torch.float64), xp=tb)
    bound.backward()
    tol_b, tol_g = (1e-9, 2e-7) if ft is np.float64 else (2e-4, 5e-3)
    assert abs(-f - float(bound.detach())) <= tol_b * abs(float(bound.detach()))
    got_A, got_b = -gfree[-2], -gfree[-1]                  # (identity transform: free state = value)
    scale = max(abs(float(A_t.grad)), abs(float(b_t.grad)))
    assert abs(got_A - float(A_t.grad)) <= tol_g * scale and abs(got_b - float(b_t.grad)) <= tol_g * scale, (got_A, got_b, A_t.grad, b_t.grad)
    r = m._resid_dev.cpu().numpy()                         # d bound / d err = d bound / d Y
    ref_r = Y_t.grad.numpy().reshape(-1)
    assert np.abs(r - ref_r).max() <= tol_g * np.abs(ref_r).max()
    if ft is np.float32:
        return
    # a fixed mean Param stays put and gets no gradient work; the free one trains
    m2 = SGPRSS(X, Y, np.sum(mk()), Z, mean_function=Constant(0.0), handle=gp_handle)
    m2.likelihood.variance = 0.3
    for k in m2.kern.kern_list:
        k.lengthscales.fixed = True
    m2.mean_function.c.fixed = True
    r_fixed = m2.optimize(maxiter=12)
    assert m2.mean_function.c.value[0] == 0.0
    m3 = SGPRSS(X, Y, np.sum(mk()), Z, mean_function=Linear(0.0, 0.0), handle=gp_handle)
    m3.likelihood.variance = 0.3
    for k in m3.kern.kern_list:
        k.lengthscales.fixed = True
    r_free = m3.optimize(maxiter=12)
    assert r_free.fun < r_fixed.fun - 1.0, (r_free.fun, r_fixed.fun)
    assert abs(m3.mean_function.b.value[0]) > 0.05 or abs(m3.mean_function.A.value[0]) > 0.05
    # the model's own bound at the trained Params is the optimiser's last objective
    assert abs(m3.build_likelihood() + r_free.fun) <= 1e-9 * abs(r_free.fun)


def test_full_cov_predictions_match_oracle(gp_handle):
    """full_cov=True of SGPR.build_predict and SGPRSS.build_predict_source (sgpr_ss.py:95-99): n x n x 1 covariances"""
    X, Y, Z, kl = _problem(600, 40, 2, 11)
    m = _model(X, Y, Z, kl, 0.3, gp_handle)
    Xs = X[::9] + 1e-5                      # n = 67: partial tiles in the n x n products
    n = Xs.shape[0]
    mean, cov = m.predict_f(Xs, full_cov=True)
    rm, rc = orc.sgpr_predict_f(Xs, X, Y, Z, kl, 0.3, full_cov=True)
    assert cov.shape == (n, n, 1)
    np.testing.assert_allclose(mean, rm, rtol=0, atol=1e-8 * np.abs(rm).max())
    np.testing.assert_allclose(cov, rc, rtol=0, atol=1e-8 * np.abs(rc).max())
    _, var = m.predict_f(Xs)
    # (K(x, x) = Kdiag * exp(-1e-6) for the Matern-1/2 envelope: euclid_dist's 1e-12 under the root)
    kd = float(np.max(orc.Kdiag_sum(kl, Xs)))
    np.testing.assert_allclose(np.diag(cov[:, :, 0]), var[:, 0], rtol=0, atol=3e-6 * kd)
    m2, c2 = m.predict_f_full_cov(Xs)
    np.testing.assert_array_equal(c2, cov)
    sm, sc = m.build_predict_source(Xs, full_cov=True)
    rsm, rsc = orc.sgpr_predict_source(Xs, X, Y, kl, 0.3, full_cov=True)
    _, sv = m.predict_s(Xs)
    for i in range(2):
        assert sc[i].shape == (n, n, 1)
        np.testing.assert_allclose(sm[i], rsm[i], rtol=0, atol=1e-8 * max(np.abs(rsm[i]).max(), 1e-3))
        np.testing.assert_allclose(sc[i], rsc[i], rtol=0, atol=1e-8 * np.abs(rsc[i]).max())
        np.testing.assert_allclose(np.diag(sc[i][:, :, 0]), sv[i][:, 0], rtol=0, atol=3e-6 * kd)


@pytest.mark.parametrize("backend,world", [("nccl", 1), ("gloo", 2)])
def test_frame_sharded_optimize_with_a_trainable_mean_function_under_a_process_group(backend, world):
    """SGPRSS(shard=...).optimize() with trainable mean-function Params all-reduces their gradient entries — a
    host array — every evaluation; on the production group (nccl only) a host tensor has no backend.  One rank through RCCL
    and two ranks through gloo (both on this box's one GPU) must run and land on the unsharded window's optimum."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29810 + (os.getpid() % 100) + world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(root, "tests", "_sharded_sgpr_child.py"), backend]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-2500:])
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
    assert len(line) == 1, r.stdout[-1500:]
    d = json.loads(line[0][7:])
    assert d["world"] == world
    assert abs(d["fun"] - d["ref_fun"]) <= 1e-8 * abs(d["ref_fun"]), d
    np.testing.assert_allclose(d["x"], d["ref_x"], rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("reg", [False, True])
def test_several_output_columns(gp_handle, reg):
    """Y of D = 3 columns (sgpr_ss.py:38 output_dim; :57-62 carry the factor D): bound, gradient (autograd through the
    oracle), predict_f / predict_s (means per column, variances tiled over the columns: :97-102), the window swap to
    another D, and a frame-sharded evaluation (emulated ranks are not needed: one rank holds every frame)"""
    X, Y1, Z, kl = _problem(1300, 48, 3, 21)
    rng = np.random.RandomState(5)
    Y = np.concatenate([Y1, 0.7 * Y1[::-1] + 0.05 * rng.randn(*Y1.shape), np.roll(Y1, 100, axis=0)], axis=1)
    m = _model(X, Y, Z, kl, 0.3, gp_handle, reg=reg)
    assert m.num_latent == 3
    got = m.build_likelihood()
    ref = orc.sgpr_bound(X, Y, Z, kl, 0.3, reg=reg)
    assert abs(got - ref) <= 1e-9 * abs(ref), (got, ref)
    ps = m._param_list()
    x0 = np.array([p.transform.backward(p.value)[0] for p in ps])
    f, gfree = m._objective(x0)
    ref_b, ref_g = _torch_bound_and_grads(X, Y, Z, kl, 0.3, reg=reg)
    assert abs(-f - ref_b) <= 1e-9 * abs(ref_b)
    np.testing.assert_allclose(-gfree * (1. + np.exp(-x0)), ref_g, rtol=0, atol=2e-7 * np.abs(ref_g).max())
    Xs = X[::11] + 1e-5
    n = Xs.shape[0]
    mean, var = m.predict_f(Xs)
    rm, rv = orc.sgpr_predict_f(Xs, X, Y, Z, kl, 0.3)
    assert mean.shape == (n, 3) and var.shape == (n, 3) and rv.shape == (n, 3)
    np.testing.assert_allclose(mean, rm, rtol=0, atol=1e-8 * np.abs(rm).max())
    np.testing.assert_allclose(var, rv, rtol=0, atol=1e-8 * np.abs(rv).max())
    _, cov = m.predict_f(Xs, full_cov=True)
    _, rc = orc.sgpr_predict_f(Xs, X, Y, Z, kl, 0.3, full_cov=True)
    assert cov.shape == (n, n, 3)
    np.testing.assert_allclose(cov, rc, rtol=0, atol=1e-8 * np.abs(rc).max())
    sm, sv = m.predict_s(Xs)
    rsm, rsv = orc.sgpr_predict_source(Xs, X, Y, kl, 0.3)
    for i in range(3):
        assert sm[i].shape == (n, 3) and sv[i].shape == (n, 3)
        np.testing.assert_allclose(sm[i], rsm[i], rtol=0, atol=1e-8 * max(np.abs(rsm[i]).max(), 1e-3))
        np.testing.assert_allclose(sv[i], rsv[i], rtol=0, atol=1e-8 * np.abs(rsv[i]).max())
    # the same model object fed a window of two columns, then one (transcription.py:253-263 swaps the DataHolders)
    m.Y = Y[:, :2]
    ref2 = orc.sgpr_bound(X, Y[:, :2], Z, kl, 0.3, reg=reg)
    assert abs(m.build_likelihood() - ref2) <= 1e-9 * abs(ref2)
    m.Y = Y[:, 2:]
    ref1 = orc.sgpr_bound(X, Y[:, 2:], Z, kl, 0.3, reg=reg)
    assert abs(m.build_likelihood() - ref1) <= 1e-9 * abs(ref1)
    # trainable mean function shared by the columns: d bound / d c against a central difference of the oracle bound
    from gpitch_amd.mean_functions import Constant
    mc = _model(X, Y, Z, kl, 0.3, gp_handle, reg=reg)
    object.__setattr__(mc, "mean_function", Constant(0.2))
    mc._compile(); mc._pack()
    psc = mc._param_list()
    xc = np.array([p.transform.backward(p.value)[0] for p in psc])
    fc, gc = mc._objective(xc)
    eps = 1e-5
    bp = orc.sgpr_bound(X, Y - (0.2 + eps), Z, kl, 0.3, reg=reg)
    bm = orc.sgpr_bound(X, Y - (0.2 - eps), Z, kl, 0.3, reg=reg)
    assert abs(-fc - orc.sgpr_bound(X, Y - 0.2, Z, kl, 0.3, reg=reg)) <= 1e-9 * abs(fc)
    assert abs(-gc[-1] - (bp - bm) / (2 * eps)) <= 1e-5 * abs(gc[-1]) + 1e-6
