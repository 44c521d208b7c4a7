"""Generate the golden vectors under tests/golden/ (run from the repo root: python tests/golden/make_golden.py).

The reference cannot be imported here (Python 2 + GPflow 0.5 + TF 1.2.1 -> ModuleNotFoundError), so these
are NOT reference outputs: they are 50-digit mpmath evaluations (oracle/mp_elbo.py) of the formulas the
oracle restates, on small seeded inputs.  They pin the numerics of the oracle and of the HIP path
(SURVEY §8c K8); parity with the reference itself stays unpinned.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import mp_elbo  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def small_problem(seed, N, M, P, m, nlin):
    rng = np.random.RandomState(seed)
    x = np.sort(rng.rand(N, 1), 0) * 3.0
    y = rng.randn(N, 1) * 0.5
    za = [(np.linspace(0.05, 2.95, M) + 0.05 * rng.rand(M)).reshape(-1, 1) for _ in range(P)]   # well separated
    zc = [(np.linspace(0.02, 2.9, M + 1) + 0.05 * rng.rand(M + 1)).reshape(-1, 1) for _ in range(P)]   # M differs per role (pdgp.py:93-94)
    ka = [{"type": "matern32", "variance": 3.5 - 0.5 * i, "lengthscales": 1.0 + 0.1 * i, "energy": [], "frequency": []}
          for i in range(P)]
    kc = [{"type": "mercer_matern12sm", "variance": 1.0 + 0.2 * i, "lengthscales": 0.1,
           "energy": list(rng.rand(m) + 0.1), "frequency": list(200. * (i + 1) * (np.arange(m) + 1.))} for i in range(P)]
    qma = [0.3 * rng.randn(M, 1) for _ in range(P)]
    qmc = [0.3 * rng.randn(M + 1, 1) for _ in range(P)]
    qsa = [(np.eye(M) + 0.05 * rng.randn(M, M))[:, :, None] for _ in range(P)]
    qsc = [(np.eye(M + 1) + 0.05 * rng.randn(M + 1, M + 1))[:, :, None] for _ in range(P)]
    return dict(x=x, y=y, za=za, zc=zc, kern_act=ka, kern_com=kc, q_mu_act=qma, q_mu_com=qmc, q_sqrt_act=qsa,
                q_sqrt_com=qsc, noise_var=0.7, N=N, P=P, nlin=nlin)


def flatten(prob):
    out = {"x": prob["x"], "y": prob["y"], "noise_var": np.array(prob["noise_var"]), "P": np.array(prob["P"]),
           "nlin": np.array(prob["nlin"])}
    for i in range(prob["P"]):
        for key in ("za", "zc", "q_mu_act", "q_mu_com", "q_sqrt_act", "q_sqrt_com"):
            out["%s_%d" % (key, i)] = prob[key][i]
        for role in ("kern_act", "kern_com"):
            k = prob[role][i]
            out["%s_%d_type" % (role, i)] = np.array(k["type"])
            out["%s_%d_theta" % (role, i)] = np.array([k["variance"], k["lengthscales"]] + list(k["energy"]) + list(k["frequency"]))
    return out


def main():
    cases = [("pdgp_small_logistic", small_problem(0, 32, 6, 2, 3, 0)),
             ("pdgp_small_softplus", small_problem(1, 24, 5, 1, 2, 1)),
             ("pdgp_small_gauss", small_problem(2, 20, 4, 3, 2, 2))]
    for name, prob in cases:
        out = flatten(prob)
        for whiten in (True, False):
            elbo, kl, fmean, fvar = mp_elbo.pdgp_elbo(prob["x"], prob["y"], prob["za"], prob["zc"], prob["kern_act"],
                                                      prob["kern_com"], prob["q_mu_act"], prob["q_sqrt_act"],
                                                      prob["q_mu_com"], prob["q_sqrt_com"], prob["noise_var"],
                                                      whiten=whiten, nlin_code=prob["nlin"], return_parts=True)
            tag = "white" if whiten else "unwhite"
            out["elbo_" + tag] = np.array(elbo)
            out["kl_" + tag] = np.array(kl)
            out["fmean_" + tag] = fmean
            out["fvar_" + tag] = fvar
        np.savez(os.path.join(HERE, name + ".npz"), **out)
        print(name, float(out["elbo_white"]), float(out["elbo_unwhite"]))
    # SGPR bound
    rng = np.random.RandomState(5)
    N, M = 40, 7
    X = np.sort(rng.rand(N, 1), 0) * 0.03
    Y = rng.randn(N, 1)
    Z = X[::6][:M].copy()
    kl_ = [{"type": "mercer_matern12sm", "variance": 1.0 + 0.3 * i, "lengthscales": 0.05 * (i + 1),
            "energy": list(rng.rand(2) + 0.2), "frequency": [150. * (i + 1), 300. * (i + 1)]} for i in range(3)]
    b = mp_elbo.sgpr_bound(X, Y, Z, kl_, 0.4)
    out = {"X": X, "Y": Y, "Z": Z, "noise_var": np.array(0.4), "bound": np.array(b), "P": np.array(3)}
    for i, k in enumerate(kl_):
        out["kern_%d_type" % i] = np.array(k["type"])
        out["kern_%d_theta" % i] = np.array([k["variance"], k["lengthscales"]] + list(k["energy"]) + list(k["frequency"]))
    np.savez(os.path.join(HERE, "sgpr_small.npz"), **out)
    print("sgpr_small", b)


if __name__ == "__main__":
    main()
