"""Exploration (GPU box): errors of the float32 strip path against the float64 oracle / float64 engine, and step times."""
import sys, os, time, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import gpitch_amd
from gpitch_amd.synth import make_problem, pdgp_from_problem
from gpitch_amd import _lib
from helpers import oracle_elbo, oracle_elbo_and_grads, model_grad_dict

def build(prob, ft):
    import gpitch_amd
    from gpitch_amd.pdgp import Pdgp
    from gpitch_amd.synth import kernels_from_problem
    kern = kernels_from_problem(prob)
    m = Pdgp(prob["x"], prob["y"], [prob["za"], prob["zc"]], kern, whiten=True, float_type=ft)
    for i in range(prob["P"]):
        m.q_mu_act[i].value = prob["q_mu_act"][i]; m.q_mu_com[i].value = prob["q_mu_com"][i]
        m.q_sqrt_act[i].value = prob["q_sqrt_act"][i]; m.q_sqrt_com[i].value = prob["q_sqrt_com"][i]
    m.likelihood.variance = prob["noise_var"]
    return m

out = {}
for (N, M, P, mpart) in [(1000, 48, 2, 3), (4200, 300, 1, 3), (8192, 256, 2, 5), (8192, 512, 1, 5)]:
    prob = make_problem(N, M, P, num_partials=mpart, seed=3)
    ref_f, ref_g = oracle_elbo_and_grads(prob)
    res = {}
    for ft in (np.float64, np.float32):
        m = build(prob, ft); m._pack()
        f = m._elbo(True)
        g = model_grad_dict(m)
        worst = {}
        for name, rg in ref_g.items():
            gg = g[name]
            if name.startswith("q_sqrt"):
                rg = np.tril(rg[:, :, 0])[:, :, None]
            sc = max(np.abs(rg).max(), 1e-12)
            worst[name.rstrip("0123456789")] = max(worst.get(name.rstrip("0123456789"), 0), float(np.abs(gg.reshape(rg.shape) - rg).max() / sc))
        res[str(np.dtype(ft))] = {"elbo_rel": abs(f - ref_f) / abs(ref_f), "grad_worst": worst}
        xs = prob["x"][::7]
        ma, va, mc, vc, ms = m.predict_act_n_com(xs)
        from oracle import gpflow05 as orc
        r = orc.pdgp_predict_act_n_com(xs, prob["za"], prob["zc"], prob["kern_act"], prob["kern_com"], prob["q_mu_act"],
                                       prob["q_sqrt_act"], prob["q_mu_com"], prob["q_sqrt_com"])
        res[str(np.dtype(ft))]["pred"] = [float(np.abs(a[0] - b[0]).max() / max(np.abs(b[0]).max(), 1e-12)) for a, b in zip((ma, va, mc, vc, ms), r)]
    out["%d_%d_%d" % (N, M, P)] = res
    print(N, M, P, json.dumps(res), flush=True)

# full-size cfg3: ELBO error and step time f64 vs f32
prob = make_problem(32768, 256, 12, num_partials=5, seed=1)
ref = float(oracle_elbo(prob))
for ft in (np.float64, np.float32):
    m = build(prob, ft)
    m.za.fixed = True; m.zc.fixed = True
    m._pack()
    f = m._elbo(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        m._elbo(True, sync=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("cfg3", np.dtype(ft), "elbo rel err", abs(f - ref) / abs(ref), "ms/eval", dt * 1e3, flush=True)
    del m

# cfg5 SGPR
from test_gpu_sgpr import _model, _problem
from oracle import gpflow05 as orc
X, Y, Z, kl = _problem(65536, 512, 5, 3)
ref = orc.sgpr_bound(X, Y, Z, kl, 0.5)
for ft in (np.float64, np.float32):
    try:
        m = _model(X, Y, Z, kl, 0.5, None, float_type=ft)
    except TypeError:
        print("test_gpu_sgpr._model has no float_type"); break
    got = m.build_likelihood()
    print("cfg5", np.dtype(ft), "bound rel err", abs(got - ref) / abs(ref), flush=True)
