"""Shared test helpers (no GPU needed to import)."""
import numpy as np


from gpitch_amd.synth import kernels_from_problem, pdgp_from_problem  # noqa: E402,F401  (product-side builders)


def oracle_elbo(prob, whiten=True, nlin_code=0, xp=None):
    from oracle import gpflow05 as orc
    from oracle.backend import NP
    xp = xp or NP
    return orc.pdgp_elbo(prob["x"], prob["y"], prob["za"], prob["zc"], prob["kern_act"], prob["kern_com"],
                         prob["q_mu_act"], prob["q_sqrt_act"], prob["q_mu_com"], prob["q_sqrt_com"],
                         prob["noise_var"], whiten=whiten, nlin_code=nlin_code, xp=xp)


def oracle_elbo_and_grads(prob, nlin_code=0, whiten=True):
    """ELBO and its gradient w.r.t. every constrained parameter by torch-CPU autograd through the
    oracle's restatement (mirrors TF reverse-mode)."""
    import torch
    from oracle import gpflow05 as orc
    from oracle.backend import TorchBackend
    tb = TorchBackend()
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)
    P = prob["P"]
    leaves = {"noise": T(prob["noise_var"])}

    def tk(d, name):
        out = dict(d)
        out["variance"] = leaves.setdefault(name + ".variance", T(d["variance"]))
        out["lengthscales"] = leaves.setdefault(name + ".lengthscales", T(d["lengthscales"]))
        out["energy"] = [leaves.setdefault("%s.energy%d" % (name, j), T(e)) for j, e in enumerate(d["energy"])]
        out["frequency"] = [leaves.setdefault("%s.frequency%d" % (name, j), T(f)) for j, f in enumerate(d["frequency"])]
        return out
    ka = [tk(d, "act%d" % i) for i, d in enumerate(prob["kern_act"])]
    kc = [tk(d, "com%d" % i) for i, d in enumerate(prob["kern_com"])]
    za = [leaves.setdefault("za%d" % i, T(prob["za"][i])) for i in range(P)]
    zc = [leaves.setdefault("zc%d" % i, T(prob["zc"][i])) for i in range(P)]
    qma = [leaves.setdefault("q_mu_act%d" % i, T(prob["q_mu_act"][i])) for i in range(P)]
    qmc = [leaves.setdefault("q_mu_com%d" % i, T(prob["q_mu_com"][i])) for i in range(P)]
    qsa = [leaves.setdefault("q_sqrt_act%d" % i, T(prob["q_sqrt_act"][i])) for i in range(P)]
    qsc = [leaves.setdefault("q_sqrt_com%d" % i, T(prob["q_sqrt_com"][i])) for i in range(P)]
    x = torch.tensor(prob["x"]); y = torch.tensor(prob["y"])
    elbo = orc.pdgp_elbo(x, y, za, zc, ka, kc, qma, qsa, qmc, qsc, leaves["noise"], whiten=whiten,
                         nlin_code=nlin_code, xp=tb)
    elbo.backward()
    return float(elbo.detach()), {k: (v.grad.numpy().copy() if v.grad is not None else None) for k, v in leaves.items()}


def model_grad_dict(m):
    """the engine's gradient vector split by parameter name (same keys as oracle_elbo_and_grads)"""
    g = m._grad.cpu().numpy()
    loc = m._local                 # pitches held by this model (all of them unless pitch-sharded)
    P = len(loc)
    out = {"noise": g[0:1].copy()}
    for gi in range(2 * P):
        act = gi < P
        i = loc[gi if act else gi - P]
        name = ("act%d" if act else "com%d") % i
        kern = (m.kern_act if act else m.kern_com)[i]
        o_th, o_z, o_mu, o_sq = m._layout[gi]
        mpart = int(kern.num_partials)
        out[name + ".variance"] = g[o_th:o_th + 1].copy()
        out[name + ".lengthscales"] = g[o_th + 1:o_th + 2].copy()
        for j in range(mpart):
            out["%s.energy%d" % (name, j)] = g[o_th + 2 + j:o_th + 3 + j].copy()
            out["%s.frequency%d" % (name, j)] = g[o_th + 2 + mpart + j:o_th + 3 + mpart + j].copy()
        M = (m.num_inducing_a if act else m.num_inducing_c)[i]
        out[("za%d" if act else "zc%d") % i] = g[o_z:o_z + M].reshape(-1, 1).copy()
        out[("q_mu_act%d" if act else "q_mu_com%d") % i] = g[o_mu:o_mu + M].reshape(-1, 1).copy()
        out[("q_sqrt_act%d" if act else "q_sqrt_com%d") % i] = g[o_sq:o_sq + M * M].reshape(M, M, 1).copy()
    return out
