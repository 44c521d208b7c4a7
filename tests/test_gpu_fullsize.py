"""BASELINE.json configurations at their full sizes on the GPU, checked against the oracle (the CPU restatement
finishes each of these in seconds to a minute) and through size-independent properties."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from helpers import pdgp_from_problem, oracle_elbo  # noqa: E402

# float64 everywhere (the fp32 configs of BASELINE.json are run at the reference's own float64: tighter, not looser)
FULLSIZE_RTOL = 1e-8


def test_cfg2_one_pitch_N32768_M512(gp_handle):
    """configs[1]: 1-pitch pdgp, N=32768, M=512, fp64 — ELBO and posterior means vs the oracle."""
    from gpitch_amd.synth import make_problem
    from oracle import gpflow05 as orc
    prob = make_problem(32768, 512, 1, num_partials=20, seed=0)
    model = pdgp_from_problem(prob, handle=gp_handle)
    got = model.compute_log_likelihood()
    ref = float(oracle_elbo(prob))
    assert abs(got - ref) <= FULLSIZE_RTOL * abs(ref), (got, ref)
    xt = prob["x"][::16].copy()
    ma, va, mc, vc, ms = model.predict_act_n_com(xt)
    r = orc.pdgp_predict_act_n_com(xt, prob["za"], prob["zc"], prob["kern_act"], prob["kern_com"], prob["q_mu_act"],
                                   prob["q_sqrt_act"], prob["q_mu_com"], prob["q_sqrt_com"])
    for got_l, ref_l in zip((ma, va, mc, vc, ms), r):
        np.testing.assert_allclose(got_l[0], ref_l[0], rtol=0, atol=1e-7 * max(np.abs(ref_l[0]).max(), 1e-3))


def test_cfg3_twelve_pitch_N32768_M256(gp_handle):
    """configs[2]: 12-pitch transcription model, N=32768, M=256 per pitch."""
    from gpitch_amd.synth import make_problem
    prob = make_problem(32768, 256, 12, num_partials=5, seed=1)
    model = pdgp_from_problem(prob, handle=gp_handle)
    got = model.compute_log_likelihood()
    ref = float(oracle_elbo(prob))
    assert abs(got - ref) <= FULLSIZE_RTOL * abs(ref), (got, ref)


def test_headline_shape_properties_N32768_M512_P12(gp_handle):
    """The bench workload (N=32768, M=512, P=12): the oracle at this size takes minutes, so check
    size-independent properties: (i) the ELBO is invariant under a permutation of the frames (what
    MinibatchData does at full batch); (ii) K1: at the prior state KL=0 and the ELBO has its closed form;
    (iii) an Adam step on the free state increases nothing pathological (finite, fvar >= 0)."""
    import gpitch_amd
    from gpitch_amd.synth import make_problem
    from oracle import gpflow05 as orc
    prob = make_problem(32768, 512, 12, num_partials=20, seed=2, trivial_q=True)
    model = pdgp_from_problem(prob, handle=gp_handle)
    e1 = model.compute_log_likelihood()      # each call draws a fresh permutation of the full batch
    e2 = model.compute_log_likelihood()
    assert abs(e1 - e2) <= 1e-11 * abs(e1)
    N, s2 = prob["N"], prob["noise_var"]
    tot = np.sum(prob["y"] ** 2)
    for i in range(prob["P"]):
        vg = prob["kern_act"][i]["variance"]
        vf = prob["kern_com"][i]["variance"] * sum(prob["kern_com"][i]["energy"])
        _, E2 = orc.hermgauss1d(np.zeros((1, 1)), np.full((1, 1), vg), 20, orc.logistic)
        tot += N * vf * E2[0, 0]
    ref = -0.5 * tot / s2 - 0.5 * N * (np.log(2 * np.pi) + np.log(s2))
    assert abs(e1 - ref) <= 1e-8 * abs(ref), (e1, ref)
    assert abs(model.build_prior_kl()) <= 1e-9
    model.za.fixed = True
    model.zc.fixed = True
    res = model.optimize(method=gpitch_amd.train.AdamOptimizer(0.0025), maxiter=2)
    assert np.isfinite(res.fun)
    ma, va = model.predict_act(prob["x"][::64])
    assert all(np.all(v > -1e-9) for v in va)


def test_cfg5_sgprss_N65536_M512_P5(gp_handle):
    """configs[4]: sgpr_ss source separation, 5 sources, N=65536, M=512 — bound vs the oracle."""
    from oracle import gpflow05 as orc
    from test_gpu_sgpr import _model, _problem
    X, Y, Z, kl = _problem(65536, 512, 5, 3)
    m = _model(X, Y, Z, kl, 0.5, gp_handle)
    got = m.build_likelihood()
    ref = orc.sgpr_bound(X, Y, Z, kl, 0.5)
    assert abs(got - ref) <= FULLSIZE_RTOL * abs(ref), (got, ref)
