"""diagnostic: ELBO / gradient reproducibility across overlap levels and repeated evaluations"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gpitch_amd.synth import make_problem, pdgp_from_problem
import gpitch_amd

def main():
    N, M, P = [int(v) for v in os.environ.get("DIAG_SHAPE", "4096,128,2").split(",")]
    levels = [int(v) for v in os.environ.get("DIAG_LEVELS", "2,2,0,0,1,2").split(",")]
    prob = make_problem(N, M, P, num_partials=3, seed=7)
    for d in prob["kern_act"]:
        d["type"] = "matern32"; d["lengthscales"] = 1.0
    h = gpitch_amd.Handle() if hasattr(gpitch_amd, "Handle") else None
    for level in levels:
        model = pdgp_from_problem(prob) if h is None else pdgp_from_problem(prob, handle=h)
        model.za.fixed = True; model.zc.fixed = True
        model._pack()
        hh = model._handle
        hh.check(hh.lib.gp_pdgp_set_overlap(model._plan, level))
        out = []
        for rep in range(3):
            f = model._elbo(True)
            g = model._grad.cpu().numpy().copy()
            out.append((f, float(np.abs(g).sum())))
        print("roles=%s level %d:" % (os.environ.get("GPITCH_AMD_SWITCHES", "default"), level), " ".join("%.15e/%.15e" % o for o in out))

main()
