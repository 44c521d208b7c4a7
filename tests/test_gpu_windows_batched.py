"""Window-batched SGPRSS (gp_sgprb_*): W windows per launch sequence, per-window parity of
the bound and its gradient with the oracle and with the one-window engine, and the batched L-BFGS-B fits against the
sequential fits of the same windows (transcription.py:265-288's loop)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import gpflow05 as orc  # noqa: E402
from test_gpu_sgpr import _model, _problem  # noqa: E402


def _windows(nwin, N, M, P, seed0=0):
    out = []
    for w in range(nwin):
        X, Y, Z, kl = _problem(N, M, P, seed0 + 17 * w)
        for p, d in enumerate(kl):                      # different hyper-parameters per window
            d["variance"] = 0.8 + 0.15 * ((w + p) % 4)
            d["lengthscales"] = 0.04 + 0.01 * ((2 * w + p) % 5)
        out.append((X + 0.125 * w, Y * (1.0 + 0.1 * w), Z + 0.125 * w, kl))
    return out


def _params_vector(noise, kl):
    v = [noise]
    for d in kl:
        v += [d["variance"], d["lengthscales"]] + list(d["energy"]) + list(d["frequency"])
    return np.array(v)


@pytest.mark.parametrize("N,M,P,reg", [(2001, 64, 3, False), (1500, 40, 2, True), (3000, 130, 2, False)])
def test_batched_bound_and_gradient_per_window(gp_handle, N, M, P, reg):
    from gpitch_amd.windows import SgprWindowBatch
    wins = _windows(5, N, M, P)
    tmpl = _model(*wins[0][:3], wins[0][3], 0.3, gp_handle, reg=reg)
    dev = SgprWindowBatch(tmpl, 6, N, M, handle=gp_handle)          # one slot more than windows: count < W
    dev.load([w[0] for w in wins], [w[1] for w in wins], [w[2] for w in wins])
    noises = [0.3 + 0.05 * i for i in range(len(wins))]
    pv = np.stack([_params_vector(nz, w[3]) for nz, w in zip(noises, wins)])
    assert pv.shape[1] == dev.nparams
    for rep in range(3):                                           # eager, captured, replayed
        bound, grad = dev.evaluate(pv)
        for i, w in enumerate(wins):
            ref = orc.sgpr_bound(w[0], w[1], w[2], w[3], noises[i], reg=reg)
            assert abs(bound[i] - ref) <= 1e-9 * abs(ref), (rep, i, bound[i], ref)
            one = _model(w[0], w[1], w[2], w[3], noises[i], gp_handle, reg=reg)
            one._compile(); one._pack()
            g1 = gp_handle.zeros(one._nparams)
            f1 = one._bound(grad=g1)
            assert abs(bound[i] - f1) <= 1e-11 * abs(f1)
            g1 = g1.cpu().numpy()
            assert np.abs(grad[i] - g1).max() <= 1e-9 * max(np.abs(g1).max(), 1e-12), (rep, i)
    # forward only
    b2, g2 = dev.evaluate(pv, with_grad=False)
    np.testing.assert_allclose(b2, bound, rtol=1e-13)
    assert g2 is None
    dev.close()


def test_fit_windows_batched_matches_sequential_fits(gp_handle):
    """the batched fits and the one-window-at-a-time fits are the same L-BFGS-B runs on the same objective"""
    from gpitch_amd.windows import fit_windows, fit_windows_batched
    wins = _windows(7, 2001, 64, 3, seed0=5)

    def make(h):
        return _model(*wins[0][:3], wins[0][3], 1.0, h)
    data = [(w[0], w[1], w[2]) for w in wins]
    seq = fit_windows(make, data, maxiter=10, num_streams=1)
    bat = fit_windows_batched(make, data, maxiter=10, batch=4)      # 7 windows in batches of 4: a partial last batch
    for a, b in zip(seq, bat):
        assert b["nfev"] >= 2 and b["nit"] <= 10
        assert abs(a["bound"] - b["bound"]) <= 1e-6 * abs(a["bound"]), (a["bound"], b["bound"])
        np.testing.assert_allclose(b["variances"], a["variances"], rtol=1e-4)
        assert abs(a["noise"] - b["noise"]) <= 1e-4 * abs(a["noise"])
    # the fits improved on the starting point and each window's result is its own
    assert len({round(b["bound"], 6) for b in bat}) == len(bat)


@pytest.mark.parametrize("N,M,P,n_new", [(700, 32, 2, None), (300, 24, 3, 211), (2001, 64, 3, None)])
def test_batched_predictions_per_window(gp_handle, N, M, P, n_new):
    """gp_sgprb_predict_f / gp_sgprb_predict_source: the per-window predictions of SoSp.optimize's loop body
    (separation.py:300-313) for W windows at once, against the one-window engine (same kernels, same order: 1e-10)
    and the oracle.  N = 300: one-workgroup factorisation; N = 700 / 2001: the blocked one, every launch batched."""
    from gpitch_amd.windows import SgprWindowBatch
    wins = _windows(4, N, M, P, seed0=3)
    tmpl = _model(*wins[0][:3], wins[0][3], 0.3, gp_handle)
    dev = SgprWindowBatch(tmpl, 5, N, M, handle=gp_handle)
    dev.load([w[0] for w in wins], [w[1] for w in wins], [w[2] for w in wins])
    noises = [0.2 + 0.05 * i for i in range(len(wins))]
    pv = np.stack([_params_vector(nz, w[3]) for nz, w in zip(noises, wins)])
    xnews = None
    if n_new is not None:
        xnews = [np.linspace(w[0].min(), w[0].max(), n_new).reshape(-1, 1) for w in wins]
    fm, fv = dev.predict_f(pv, xnews)
    sm, sv = dev.predict_s(pv, xnews, chunk=3)                 # 4 windows in chunks of 3: a partial second chunk
    n = N if n_new is None else n_new
    assert fm.shape == (4, n) and sm.shape == (4, P, n)
    for i, w in enumerate(wins):
        xs = w[0] if xnews is None else xnews[i]
        one = _model(w[0], w[1], w[2], w[3], noises[i], gp_handle)
        m1, v1 = one.predict_f(xs)
        np.testing.assert_allclose(fm[i], m1[:, 0], rtol=0, atol=1e-10 * np.abs(m1).max())
        np.testing.assert_allclose(fv[i], v1[:, 0], rtol=0, atol=1e-10 * np.abs(v1).max())
        ms, vs = one.predict_s(xs)
        for k in range(P):
            np.testing.assert_allclose(sm[i, k], ms[k][:, 0], rtol=0, atol=1e-10 * max(np.abs(ms[k]).max(), 1e-12))
            np.testing.assert_allclose(sv[i, k], vs[k][:, 0], rtol=0, atol=1e-10 * np.abs(vs[k]).max())
    # one window against the oracle (the small cases: the oracle's N x N factorisation on the host)
    if N <= 700:
        w = wins[1]
        xs = w[0] if xnews is None else xnews[1]
        rm, rv = orc.sgpr_predict_f(xs, w[0], w[1], w[2], w[3], noises[1])
        assert np.abs(fm[1] - rm[:, 0]).max() <= 1e-8 * np.abs(rm).max()
        assert np.abs(fv[1] - rv[:, 0]).max() <= 1e-8 * np.abs(rv).max()
        rms, rvs = orc.sgpr_predict_source(xs, w[0], w[1], w[3], noises[1])
        for k in range(P):
            assert np.abs(sm[1, k] - np.asarray(rms[k]).reshape(-1)).max() <= 1e-7 * max(np.abs(rms[k]).max(), 1e-12)
            assert np.abs(sv[1, k] - np.asarray(rvs[k]).reshape(-1)).max() <= 1e-7 * np.abs(rvs[k]).max()
    # the predictions leave the plan usable: the bound afterwards is the bound before
    b, _ = dev.evaluate(pv, with_grad=False)
    for i, w in enumerate(wins):
        ref = orc.sgpr_bound(w[0], w[1], w[2], w[3], noises[i])
        assert abs(b[i] - ref) <= 1e-9 * abs(ref)
    dev.close()


def test_fit_windows_batched_with_predictions(gp_handle):
    """fit + predict_f + predict_s for every window, as SoSp.optimize does (separation.py:279-313)"""
    from gpitch_amd.windows import fit_windows_batched
    wins = _windows(5, 801, 32, 2, seed0=9)

    def make(h):
        return _model(*wins[0][:3], wins[0][3], 1.0, h)
    data = [(w[0], w[1], w[2]) for w in wins]
    res = fit_windows_batched(make, data, maxiter=5, batch=3, predict=True)
    for r, w in zip(res, wins):
        assert r["mean"].shape == (801, 1) and r["var"].shape == (801, 1)
        assert len(r["smean"]) == 2 and r["smean"][0].shape == (801, 1) and len(r["svar"]) == 2
        assert np.all(r["var"] > 0) and all(np.all(v > -1e-9) for v in r["svar"])
        # the same numbers from a one-window model put at the fitted parameters
        one = make(gp_handle)
        one.X, one.Y, one.Z = w[0], w[1], w[2]
        pv = r["params"]
        one.likelihood.variance = pv[0]
        o = 1
        for k in one.kern.kern_list:
            m = int(k.num_partials)
            k.variance, k.lengthscales = pv[o], pv[o + 1]
            for q in range(m):
                k.energy[q].value, k.frequency[q].value = pv[o + 2 + q], pv[o + 2 + m + q]
            o += 2 + 2 * m
        m1, v1 = one.predict_f(w[0])
        np.testing.assert_allclose(r["mean"], m1, rtol=0, atol=1e-9 * np.abs(m1).max())
        np.testing.assert_allclose(r["var"], v1, rtol=0, atol=1e-9 * np.abs(v1).max())
        ms, vs = one.predict_s(w[0])
        np.testing.assert_allclose(r["smean"][1], ms[1], rtol=0, atol=1e-9 * max(np.abs(ms[1]).max(), 1e-12))
        np.testing.assert_allclose(r["svar"][0], vs[0], rtol=0, atol=1e-9 * np.abs(vs[0]).max())


def test_batched_prediction_argument_and_failure_paths(gp_handle):
    """status codes of the window-batched predictions: bad sizes are GP_ERR_BAD_ARG, a workspace that is too small
    GP_ERR_WORKSPACE, a window whose exact-GP covariance is not positive definite GP_ERR_NOT_PD (raised, not a wrong
    number) — and the plan stays usable afterwards"""
    from gpitch_amd import _lib
    from gpitch_amd.windows import SgprWindowBatch
    N, M, P = 600, 24, 2
    wins = _windows(2, N, M, P, seed0=4)
    tmpl = _model(*wins[0][:3], wins[0][3], 0.3, gp_handle)
    dev = SgprWindowBatch(tmpl, 2, N, M, handle=gp_handle)
    dev.load([w[0] for w in wins], [w[1] for w in wins], [w[2] for w in wins])
    pv = np.stack([_params_vector(0.3, w[3]) for w in wins])
    h = gp_handle
    with pytest.raises(ValueError):                                     # more new points than frames per window
        dev.predict_f(pv, [np.linspace(0, 1, N + 5).reshape(-1, 1)] * 2)
    mean, var = h.empty(2, N), h.empty(2, N)
    st = h.lib.gp_sgprb_predict_f(dev.plan, dev.params.data_ptr(), dev.X.data_ptr(), dev.Y.data_ptr(), dev.Z.data_ptr(),
                                  dev.X.data_ptr(), N, 3, mean.data_ptr(), var.data_ptr())       # count > windows of the plan
    assert st == _lib.GP_ERR_BAD_ARG
    small = h.workspace(1024)
    sm, sv = h.empty(2, P, N), h.empty(2, P, N)
    st = h.lib.gp_sgprb_predict_source(dev.plan, dev.params.data_ptr(), dev.X.data_ptr(), dev.Y.data_ptr(), dev.X.data_ptr(),
                                       N, 2, sm.data_ptr(), sv.data_ptr(), small.data_ptr(), small.numel())
    assert st == _lib.GP_ERR_WORKSPACE
    bad = pv.copy()
    bad[1, 0] = -50.0                                                   # noise variance: K + s2 I loses definiteness
    with pytest.raises(_lib.NotPositiveDefiniteError):
        dev.predict_s(bad)
    fm, fv = dev.predict_f(pv)                                          # still in working order
    one = _model(wins[1][0], wins[1][1], wins[1][2], wins[1][3], 0.3, gp_handle)
    m1, v1 = one.predict_f(wins[1][0])
    np.testing.assert_allclose(fm[1], m1[:, 0], rtol=0, atol=1e-10 * np.abs(m1).max())
    dev.close()


def test_a_window_that_turns_not_positive_definite_mid_fit_is_retired_alone(gp_handle, monkeypatch):
    """a failed Cholesky inside a batched fit must not hand finite garbage to L-BFGS-B.  The evaluation's own
    status word comes down with its results; the failing window is retired with results[i]["error"], the others finish
    exactly as in a clean run, the flag does not leak to the next user of the handle, and predict=True still works."""
    from gpitch_amd import windows as W
    wins = _windows(6, 801, 32, 2, seed0=21)
    data = [(w[0], w[1], w[2]) for w in wins]

    def make(h):
        return _model(*wins[0][:3], wins[0][3], 1.0, h)
    clean = W.fit_windows_batched(make, data, maxiter=6, batch=6, handle=None, predict=True, inflight=1)
    assert all("error" not in r for r in clean)
    real_submit = W.SgprWindowBatch.submit
    calls = {"n": 0}

    def bad_submit(self, params_host, with_grad=True):
        calls["n"] += 1
        p = np.array(params_host, dtype=np.float64, copy=True)
        if calls["n"] == 3:                    # the third evaluation of the fit: window slot 4 gets a negative kernel variance
            p[4, 1] = -3.0                     # -> its Kuu is negative definite
        return real_submit(self, p, with_grad)
    monkeypatch.setattr(W.SgprWindowBatch, "submit", bad_submit)
    res = W.fit_windows_batched(make, data, maxiter=6, batch=6, handle=None, predict=True, inflight=1)
    monkeypatch.setattr(W.SgprWindowBatch, "submit", real_submit)
    assert "error" in res[4] and "positive definite" in res[4]["error"] and np.isnan(res[4]["bound"])
    assert "mean" not in res[4]
    for i in (0, 1, 2, 3, 5):
        assert "error" not in res[i]
        assert res[i]["bound"] == clean[i]["bound"] and res[i]["nfev"] == clean[i]["nfev"]
        np.testing.assert_array_equal(res[i]["params"], clean[i]["params"])
        np.testing.assert_array_equal(res[i]["mean"], clean[i]["mean"])
    # nothing left behind on the default handle either: a Pdgp Adam loop on it would freeze at step 1 otherwise
    gp_handle.check(gp_handle.lib.gp_check_not_pd(gp_handle.h))


def test_a_window_not_positive_definite_at_its_starting_parameters_is_retired_alone(gp_handle, monkeypatch):
    """a window whose Kuu is not positive definite at its STARTING parameters fails in every evaluation it takes
    part in with its own parameters; retired, it rides along on a healthy window's parameters, so the repeat (and every later
    evaluation) goes through and the other windows finish exactly as in a clean run."""
    from gpitch_amd import windows as W
    wins = _windows(5, 801, 32, 2, seed0=33)
    data = [(w[0], w[1], w[2]) for w in wins]

    def make(h):
        return _model(*wins[0][:3], wins[0][3], 1.0, h)
    clean = W.fit_windows_batched(make, data, maxiter=5, batch=5, handle=None, inflight=1)
    real_submit = W.SgprWindowBatch.submit

    def bad_submit(self, params_host, with_grad=True):
        p = np.array(params_host, dtype=np.float64, copy=True)
        if not any(np.array_equal(p[2], p[q]) for q in (0, 1, 3, 4)):      # slot 2 still on parameters of its own
            p[2, 1] = -3.0                                                   # -> negative kernel variance: Kuu not PD
        return real_submit(self, p, with_grad)
    monkeypatch.setattr(W.SgprWindowBatch, "submit", bad_submit)
    res = W.fit_windows_batched(make, data, maxiter=5, batch=5, handle=None, inflight=1)
    monkeypatch.setattr(W.SgprWindowBatch, "submit", real_submit)
    assert "error" in res[2] and np.isnan(res[2]["bound"])
    for i in (0, 1, 3, 4):
        assert "error" not in res[i]
        assert res[i]["bound"] == clean[i]["bound"] and res[i]["nfev"] == clean[i]["nfev"]
        np.testing.assert_array_equal(res[i]["params"], clean[i]["params"])
    gp_handle.check(gp_handle.lib.gp_check_not_pd(gp_handle.h))


def test_evaluate_raises_for_a_bad_window_and_a_new_bound_buffer_is_not_served_from_the_old_graph(gp_handle):
    """(i) SgprWindowBatch.evaluate reports a failed Cholesky instead of returning numbers; (ii) ADVICE r2: predict_f runs
    the forward pass with its own bound buffer — a following evaluate() with the same window count must write self.bound,
    not replay the graph captured for predict_f's address."""
    from gpitch_amd import _lib
    from gpitch_amd.windows import SgprWindowBatch
    import torch
    N, M, P = 600, 24, 2
    wins = _windows(3, N, M, P, seed0=8)
    s = torch.cuda.Stream(device=gp_handle.device)
    with torch.cuda.stream(s):
        h = _lib.Handle(gp_handle.device.index, stream=s)         # a stream of its own: graphs are recorded
        tmpl = _model(*wins[0][:3], wins[0][3], 0.3, h)
        dev = SgprWindowBatch(tmpl, 3, N, M, handle=h)
        dev.load([w[0] for w in wins], [w[1] for w in wins], [w[2] for w in wins])
        pv = np.stack([_params_vector(0.3, w[3]) for w in wins])
        bad = pv.copy()
        bad[2, 1] = -3.0
        with pytest.raises(_lib.NotPositiveDefiniteError, match="window 2"):
            dev.evaluate(bad)
        b0, _ = dev.evaluate(pv, with_grad=False)                 # the flag was cleared with the failing evaluation
        for _ in range(3):                                        # eager, captured, replayed — at predict_f's bound address
            dev.predict_f(pv)
        pv2 = pv.copy()
        pv2[:, 0] = 0.45
        b1, _ = dev.evaluate(pv2, with_grad=False)
        for i, w in enumerate(wins):
            ref = orc.sgpr_bound(w[0], w[1], w[2], w[3], 0.45)
            assert abs(b1[i] - ref) <= 1e-9 * abs(ref), (i, b1[i], ref, b0[i])
        dev.close()
        tmpl._destroy()
        s.synchronize()
        h.close()


def test_fit_windows_batched_refuses_what_it_would_silently_drop(gp_handle):
    from gpitch_amd import mean_functions
    from gpitch_amd.windows import fit_windows_batched
    wins = _windows(2, 400, 16, 2, seed0=2)
    data = [(w[0], w[1], w[2]) for w in wins]
    from gpitch_amd.sgpr_ss import SGPRSS
    from gpitch_amd.synth import kernels_from_problem

    def make_mf(h):
        m = _model(*wins[0][:3], wins[0][3], 1.0, h)
        object.__setattr__(m, "mean_function", mean_functions.Constant(0.1))
        return m
    with pytest.raises(NotImplementedError, match="mean_function"):
        fit_windows_batched(make_mf, data, maxiter=2, batch=2)

    def make_f32(h):
        m = _model(*wins[0][:3], wins[0][3], 1.0, h)
        m._bits = 32
        return m
    with pytest.raises(NotImplementedError, match="float64"):
        fit_windows_batched(make_f32, data, maxiter=2, batch=2)
