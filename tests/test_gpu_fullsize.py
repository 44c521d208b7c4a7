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


def _k1_closed_form(y, N, kern_act, kern_com, s2):
    """K1: at q_mu = 0, q_sqrt = I (whitened) KL = 0, fmean = 0, fvar = Kdiag, so
    ELBO = -1/2 [sum y^2 + N sum_i v_f,i E2_i] / s2 - N/2 (log 2 pi + log s2), independent of M, Z and the lengthscales."""
    from oracle import gpflow05 as orc
    tot = np.sum(y ** 2)
    for ka, kc in zip(kern_act, kern_com):
        _, E2 = orc.hermgauss1d(np.zeros((1, 1)), np.full((1, 1), ka["variance"]), 20, orc.logistic)
        tot += N * kc["variance"] * sum(kc["energy"]) * E2[0, 0]
    return -0.5 * tot / s2 - 0.5 * N * (np.log(2 * np.pi) + np.log(s2))


def test_cfg5_as_pdgp_five_sources_N65536_M512(gp_handle):
    """configs[4] reads "5 sources x activation + component GPs", which sgpr_ss does not have: the same
    size as a Pdgp with 5 pitches (10 latent GPs), float64 and float32 strips against the oracle"""
    from gpitch_amd.synth import make_problem
    prob = make_problem(65536, 512, 5, num_partials=5, seed=2)
    ref = float(oracle_elbo(prob))
    model = pdgp_from_problem(prob, handle=gp_handle)
    got = model.compute_log_likelihood()
    assert abs(got - ref) <= FULLSIZE_RTOL * abs(ref), (got, ref)
    model = None
    m32 = pdgp_from_problem(prob, handle=gp_handle, float_type=np.float32)
    got32 = m32.compute_log_likelihood()
    assert abs(got32 - ref) <= 2e-4 * abs(ref), (got32, ref)       # the stated float32 tolerance (tests/test_gpu_f32.py)


def test_cfg4_windowed_N262144_twelve_pitches(gp_handle):
    """configs[3]: 12 pitches, N = 262144 'windowed', one window per GPU on 8 GPUs.  The reference cuts long audio
    into independent pieces that are fitted one after another (window_overlap.py:194-211 non-overlapping segments;
    :7-16 half-overlapping windows of 2001; transcription.py:265-288): here the 8 x 32768 segments go through
    dist.window_assignment exactly as 8 ranks would take them (run one after another on the one GPU of this box),
    every window at full size (M = 512, P = 12) against the K1 closed form, the job's total = the sum the scalar
    all-reduce forms, and one window at the transcription model's size (M = 256) against the oracle."""
    import gpitch_amd
    from gpitch_amd import dist as gp_dist
    from gpitch_amd.synth import make_problem, uniform_inducing
    from gpitch_amd.window_overlap import segmented, windowed
    NT, NW, P = 262144, 32768, 12
    long_prob = make_problem(NT, 512, P, num_partials=20, seed=4, trivial_q=True)
    xs, ys = segmented(long_prob["x"], long_prob["y"], window_size=NW)
    assert len(xs) == 8 and all(xi.shape == (NW, 1) for xi in xs)
    xw, yw = windowed(long_prob["x"], long_prob["y"], 2001)
    assert len(xw) == 261 and xw[-1].shape == (2001, 1)          # the other split SURVEY section 8d names

    def window_problem(w, M, trivial=True, seed=0):
        p = make_problem(NW, M, P, num_partials=20 if trivial else 5, seed=seed, trivial_q=trivial)
        p["x"], p["y"] = xs[w].copy(), ys[w].copy()
        z = uniform_inducing(p["x"], M)
        p["za"], p["zc"] = [z.copy() for _ in range(P)], [z.copy() for _ in range(P)]
        return p
    world = 8
    seen, total, total_ref = [], 0.0, 0.0
    for rank in range(world):
        for w in gp_dist.window_assignment(len(xs), world, rank):
            prob = window_problem(w, 512)
            model = pdgp_from_problem(prob, handle=gp_handle)
            got = model.compute_log_likelihood()
            ref = _k1_closed_form(prob["y"], NW, prob["kern_act"], prob["kern_com"], prob["noise_var"])
            assert abs(got - ref) <= 1e-8 * abs(ref), (w, got, ref)
            total += got
            total_ref += ref
            seen.append(w)
            del model
    assert sorted(seen) == list(range(8))                         # every window fitted exactly once
    assert abs(total - total_ref) <= 1e-8 * abs(total_ref)
    # one window (the sixth) with a non-trivial variational state, at the transcription model's M, against the oracle
    prob = window_problem(5, 256, trivial=False, seed=9)
    model = pdgp_from_problem(prob, handle=gp_handle)
    got = model.compute_log_likelihood()
    ref = float(oracle_elbo(prob))
    assert abs(got - ref) <= FULLSIZE_RTOL * abs(ref), (got, ref)


def test_headline_launch_against_the_oracle_N32768_M512_P12_m20(gp_handle):
    """The workload bench.py times (make_problem(32768, 512, 12, num_partials=20, seed=0): 4 x 4 tile grid x 24 latent GPs in
    every strip launch, NON-trivial q_mu / q_sqrt so both big products contribute): forward ELBO against the oracle's numpy
    pass over the same arrays, and one pitch's posterior moments / source mean (pdgp.py:190-208) against the oracle's
    conditionals for that pitch."""
    from gpitch_amd.synth import make_problem
    from oracle import gpflow05 as orc
    prob = make_problem(32768, 512, 12, num_partials=20, seed=0)
    model = pdgp_from_problem(prob, handle=gp_handle)
    got = model.compute_log_likelihood()
    ref = float(oracle_elbo(prob))
    print("headline ELBO: HIP %.12e oracle %.12e relative difference %.2e" % (got, ref, abs(got / ref - 1)))
    assert abs(got - ref) <= FULLSIZE_RTOL * abs(ref), (got, ref)
    p = 7
    xt = prob["x"][::8].copy()
    ma, va, mc, vc, ms = model.predict_act_n_com(xt)
    r = orc.pdgp_predict_act_n_com(xt, [prob["za"][p]], [prob["zc"][p]], [prob["kern_act"][p]], [prob["kern_com"][p]],
                                   [prob["q_mu_act"][p]], [prob["q_sqrt_act"][p]], [prob["q_mu_com"][p]],
                                   [prob["q_sqrt_com"][p]])
    for got_l, ref_l in zip((ma, va, mc, vc, ms), r):
        np.testing.assert_allclose(got_l[p], ref_l[0], rtol=0, atol=1e-7 * max(np.abs(ref_l[0]).max(), 1e-3))


def test_headline_backward_launch_against_autograd_N32768_M512_P12_m20(gp_handle):
    """The BACKWARD pass of the launch bench.py times (make_problem(32768, 512, 12, num_partials=20, seed=0), inducing
    inputs fixed as in demo-modgp.py:40-41 = the bench's state): every gradient block — noise, the 24 kernels' variance /
    lengthscale / energies / frequencies, q_mu and tril(q_sqrt) of the 24 latent GPs — against torch autograd through the
    oracle (TF reverse mode of pdgp.py:133-170) at the full 4 x 4-tile x 24-GP grid, which is the shape at which the
    split-K product, the dense Kuf_bar product, the contraction fused into its epilogue (stationary family) and the
    row-streaming contraction (spectral-mixture family) run in the benchmark.  Bound: 2e-5 of each block's largest entry
    (cond(Kuu) ~ 1e9 on the activation side: both sides carry cond * eps), as at M = 512, P = 1 in test_gpu_pdgp.py."""
    from gpitch_amd.synth import make_problem
    from helpers import oracle_elbo_and_grads, model_grad_dict
    prob = make_problem(32768, 512, 12, num_partials=20, seed=0)
    model = pdgp_from_problem(prob, handle=gp_handle)
    model.za.fixed = True
    model.zc.fixed = True
    model._pack()
    f = model._elbo(True)
    got_g = model_grad_dict(model)
    ref_f, ref_g = oracle_elbo_and_grads(prob)
    print("headline ELBO (gradient pass): HIP %.12e oracle %.12e" % (f, ref_f))
    assert abs(f - ref_f) <= FULLSIZE_RTOL * abs(ref_f), (f, ref_f)
    worst = {}
    for name, rg in ref_g.items():
        if name.startswith("za") or name.startswith("zc"):
            continue                                   # fixed: not part of the step
        gg = got_g[name]
        if name.startswith("q_sqrt"):
            rg = np.tril(rg[:, :, 0])[:, :, None]
            assert np.all(np.triu(gg[:, :, 0], 1) == 0)
        scale = max(np.abs(rg).max(), 1e-12)
        worst[name] = np.abs(gg.reshape(rg.shape) - rg).max() / scale
    by_kind = {}
    for k, v in worst.items():
        kind = k.rstrip("0123456789").split(".")[-1] + ("(act)" if "act" in k else "(com)" if "com" in k else "")
        by_kind[kind] = max(by_kind.get(kind, 0.0), v)
    print("headline gradient, worst relative deviation per block kind:", {k: "%.1e" % v for k, v in sorted(by_kind.items())})
    bad = {k: v for k, v in worst.items() if v > 2e-5}
    assert not bad, bad
