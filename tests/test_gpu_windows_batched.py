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
