"""Fused (z fixed) against separate (z free) Kuf-side contraction of the stationary family, float32 strips, at cfg3 shapes:
the activation kernels' gradients of the two models may differ by summation order only.  python tools/check_fused_contraction.py"""
import numpy as np, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import model_grad_dict
from gpitch_amd.pdgp import Pdgp
from gpitch_amd.synth import make_problem, kernels_from_problem
from gpitch_amd import _lib
h = _lib.default_handle()
for (N, M, P, m) in ((8192, 256, 3, 5), (32768, 256, 12, 5)):
    prob = make_problem(N, M, P, num_partials=m, seed=1)
    out = []
    for fixed in (True, False):
        mdl = Pdgp(prob["x"], prob["y"], [prob["za"], prob["zc"]], kernels_from_problem(prob), handle=h, float_type=np.float32)
        for i in range(P):
            mdl.q_mu_act[i].value = prob["q_mu_act"][i]; mdl.q_mu_com[i].value = prob["q_mu_com"][i]
            mdl.q_sqrt_act[i].value = prob["q_sqrt_act"][i]; mdl.q_sqrt_com[i].value = prob["q_sqrt_com"][i]
        mdl.likelihood.variance = prob["noise_var"]
        if fixed:
            mdl.za.fixed = True; mdl.zc.fixed = True
        mdl._pack()
        f = mdl._elbo(True)
        g = model_grad_dict(mdl)
        out.append((f, {k: v.copy() for k, v in g.items() if k.startswith("act")}))
        del mdl
    print(N, M, P, "elbo", out[0][0], out[1][0])
    for k in sorted(out[0][1]):
        a, b = out[0][1][k], out[1][1][k]
        print("  %-22s fused %+.12e separate %+.12e rel %.2e" % (k, a.ravel()[0], b.ravel()[0], abs(a - b).max() / max(abs(b).max(), 1e-300)))
