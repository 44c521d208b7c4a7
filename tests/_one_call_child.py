"""child of tests/test_gpu_bench.py::test_one_call_sharded_steps_match_the_two_stage_path (its own process: the RCCL
communicator lives on the process's default handle)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import gpitch_amd
    from gpitch_amd.synth import make_problem, pdgp_from_problem
    from gpitch_amd.train import AdamOptimizer
    prob = make_problem(4096, 64, 3, num_partials=3, seed=5)
    for shard in ((0, 1), ("gp", 0, 1)):
        out = []
        for one_call in (True, False):
            m = pdgp_from_problem(prob, shard=shard)
            m.za.fixed = True
            m.zc.fixed = True
            m._pack()
            if not one_call:
                object.__setattr__(m, "_comm_cache", None)          # the two-stage path, exchange by the caller
            else:
                assert m._sharded_comm() is not None, "no RCCL communicator on this box"
            f = m._elbo(True)
            g = m._grad.cpu().numpy().copy()
            res = m.optimize(method=AdamOptimizer(0.005), maxiter=3)
            out.append((f, g, res.fun, np.array(res.x)))
        (f1, g1, r1, x1), (f0, g0, r0, x0) = out
        assert f1 == f0 and np.array_equal(g1, g0), (shard, f1, f0, np.abs(g1 - g0).max())
        assert r1 == r0 and np.array_equal(x1, x0), (shard, r1, r0)
        print("shard", shard, "ELBO", f1, "after 3 Adam steps", r1)
    # frame-sharded SGPRSS, one rank
    from test_gpu_sgpr import _model, _problem
    X, Y, Z, kl = _problem(3000, 64, 2, 7)
    vals = []
    for one_call in (True, False):
        from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
        from gpitch_amd.sgpr_ss import SGPRSS
        ks = [MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=d["variance"],
                               lengthscales=d["lengthscales"]) for d in kl]
        m = SGPRSS(X, Y, np.sum(ks), Z, shard=(0, 1))
        m.likelihood.variance = 0.3
        if not one_call:
            object.__setattr__(m, "_comm_cache", None)
        res = m.optimize(maxiter=3)
        vals.append((res.fun, np.array(res.x)))
    assert vals[0][0] == vals[1][0] and np.array_equal(vals[0][1], vals[1][1]), vals
    print("ONE-CALL OK")


if __name__ == "__main__":
    main()
