"""per-block gradient deviation of the float32 strip path from the float64 oracle (tests/test_gpu_f32.py's measure), for
one shape: F32_SHAPE=N,M,P,m.  Run under different GPITCH_AMD_SWITCHES to compare kernel forms."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from helpers import oracle_elbo_and_grads, model_grad_dict
from gpitch_amd._lib import Handle
from gpitch_amd.pdgp import Pdgp
from gpitch_amd.synth import make_problem, kernels_from_problem

N, M, P, m = (int(v) for v in os.environ.get("F32_SHAPE", "2048,128,2,3").split(","))
prob = make_problem(N, M, P, num_partials=m, seed=int(os.environ.get("F32_SEED", "3")))
h = Handle(0)
model = Pdgp(prob["x"], prob["y"], [prob["za"], prob["zc"]], kernels_from_problem(prob), whiten=True, handle=h, float_type=np.float32)
for i in range(P):
    model.q_mu_act[i].value = prob["q_mu_act"][i]; model.q_mu_com[i].value = prob["q_mu_com"][i]
    model.q_sqrt_act[i].value = prob["q_sqrt_act"][i]; model.q_sqrt_com[i].value = prob["q_sqrt_com"][i]
model.likelihood.variance = prob["noise_var"]
model._pack()
f = model._elbo(True)
ref_f, ref_g = oracle_elbo_and_grads(prob)
got = model_grad_dict(model)
out = {}
for name, rg in ref_g.items():
    gg = got[name]
    if name.startswith("q_sqrt"):
        rg = np.tril(rg[:, :, 0])[:, :, None]
    out[name] = float(np.abs(gg.reshape(rg.shape) - rg).max() / max(np.abs(rg).max(), 1e-12))
print(os.environ.get("GPITCH_AMD_SWITCHES", ""), "elbo rel", abs(f - ref_f) / abs(ref_f))
print("  ", {k: float("%.2e" % v) for k, v in sorted(out.items(), key=lambda kv: -kv[1])[:8]})
