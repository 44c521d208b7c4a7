"""The oracle against the committed golden vectors (50-digit mpmath; tests/golden/make_golden.py).
CPU only.  The goldens pin numerics, not reference semantics: parity with the reference is UNPINNED
(it ships no tests/fixtures and cannot be imported: Python 2 + GPflow 0.5 + TF 1.2.1)."""
import os

import numpy as np
import pytest

from oracle import gpflow05 as orc

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_pdgp(name):
    d = np.load(os.path.join(HERE, name + ".npz"))
    P = int(d["P"])

    def kern(role, i):
        th = d["%s_%d_theta" % (role, i)]
        t = str(d["%s_%d_type" % (role, i)])
        m = (len(th) - 2) // 2
        return {"type": t, "variance": th[0], "lengthscales": th[1], "energy": list(th[2:2 + m]),
                "frequency": list(th[2 + m:])}
    prob = dict(x=d["x"], y=d["y"], noise_var=float(d["noise_var"]), P=P, nlin=int(d["nlin"]),
                za=[d["za_%d" % i] for i in range(P)], zc=[d["zc_%d" % i] for i in range(P)],
                kern_act=[kern("kern_act", i) for i in range(P)], kern_com=[kern("kern_com", i) for i in range(P)],
                q_mu_act=[d["q_mu_act_%d" % i] for i in range(P)], q_mu_com=[d["q_mu_com_%d" % i] for i in range(P)],
                q_sqrt_act=[d["q_sqrt_act_%d" % i] for i in range(P)], q_sqrt_com=[d["q_sqrt_com_%d" % i] for i in range(P)])
    return prob, d


PDGP_CASES = ["pdgp_small_logistic", "pdgp_small_softplus", "pdgp_small_gauss"]


@pytest.mark.parametrize("name", PDGP_CASES)
@pytest.mark.parametrize("whiten", [True, False])
def test_oracle_matches_mpmath_golden(name, whiten):
    prob, d = load_pdgp(name)
    tag = "white" if whiten else "unwhite"
    elbo = orc.pdgp_elbo(prob["x"], prob["y"], prob["za"], prob["zc"], prob["kern_act"], prob["kern_com"],
                         prob["q_mu_act"], prob["q_sqrt_act"], prob["q_mu_com"], prob["q_sqrt_com"], prob["noise_var"],
                         whiten=whiten, nlin_code=prob["nlin"])
    assert abs(elbo - float(d["elbo_" + tag])) <= 1e-11 * abs(float(d["elbo_" + tag]))
    fmean, fvar = orc.pdgp_conditionals(prob["x"], prob["za"], prob["zc"], prob["kern_act"], prob["kern_com"],
                                        prob["q_mu_act"], prob["q_sqrt_act"], prob["q_mu_com"], prob["q_sqrt_com"], whiten)
    np.testing.assert_allclose(fmean, d["fmean_" + tag], rtol=0, atol=1e-11 * np.abs(d["fmean_" + tag]).max())
    np.testing.assert_allclose(fvar, d["fvar_" + tag], rtol=0, atol=1e-11 * np.abs(d["fvar_" + tag]).max())
    kl = orc.pdgp_prior_kl(prob["za"], prob["zc"], prob["kern_act"], prob["kern_com"], prob["q_mu_act"],
                           prob["q_sqrt_act"], prob["q_mu_com"], prob["q_sqrt_com"], whiten)
    assert abs(kl - float(d["kl_" + tag])) <= 1e-11 * abs(float(d["kl_" + tag]))


def test_oracle_sgpr_bound_matches_golden():
    d = np.load(os.path.join(HERE, "sgpr_small.npz"))
    kl = []
    for i in range(int(d["P"])):
        th = d["kern_%d_theta" % i]
        m = (len(th) - 2) // 2
        kl.append({"type": str(d["kern_%d_type" % i]), "variance": th[0], "lengthscales": th[1],
                   "energy": list(th[2:2 + m]), "frequency": list(th[2 + m:])})
    b = orc.sgpr_bound(d["X"], d["Y"], d["Z"], kl, float(d["noise_var"]))
    assert abs(b - float(d["bound"])) <= 1e-11 * abs(float(d["bound"]))


def test_torch_backend_agrees_with_numpy():
    import torch
    from oracle.backend import TorchBackend
    prob, d = load_pdgp("pdgp_small_logistic")
    tb = TorchBackend()
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float64))
    e = orc.pdgp_elbo(T(prob["x"]), T(prob["y"]), [T(z) for z in prob["za"]], [T(z) for z in prob["zc"]],
                      prob["kern_act"], prob["kern_com"], [T(a) for a in prob["q_mu_act"]],
                      [T(a) for a in prob["q_sqrt_act"]], [T(a) for a in prob["q_mu_com"]],
                      [T(a) for a in prob["q_sqrt_com"]], T(prob["noise_var"]), whiten=True, nlin_code=0, xp=tb)
    assert abs(float(e) - float(d["elbo_white"])) <= 1e-11 * abs(float(d["elbo_white"]))
