"""Generates tests/golden/demo_real_audio_anchor.npz — the reference's printed real-audio ELBO reproduced by the oracle.

The reference holds exactly one printed result of a fully deterministic ELBO job:
/root/reference/demos/notebooks/demo_modgp-real-audio.ipynb cells 4-9 (shipped recording, init_liv -> 109 inducing
points, Matern32 + MercerMatern12sm(5 partials), minibatch 100 from RandomState(0), Adam(0.0025) x 10000 with a logger
that evaluates the objective on every 10th iteration) prints
    fun: -69632.62624963776,  jac: [-590.866639, 4250.27187, 657.101668, ..., 0.0410200613, 0.265353723, -0.703963363],
    x:   [0.93966396, 1.04036735, 3.99887154, ..., 0.07770003, 0.31399524, 0.6851173]
oracle/demo_anchor.py restates that job (the recording's samples come from tests/golden/init_liv_real_audio.npz: data,
not code; nothing under /root/reference is read).  This script runs it on the CPU (about 2.5 minutes, one thread) and
stores the oracle's result next to the printed values, plus what the GPU test needs to compare the HIP path with the
oracle without re-running it: snapshots of the free state along the trajectory, the logger's trace, the final
constrained parameters and the oracle's predictions (pdgp.py:190-208) at x[::3] (cell 11) at that final state.

    python tests/golden/make_demo_anchor.py
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

SNAP_AT = (10, 100, 500, 1000, 2000, 5000)


def main():
    import torch
    torch.set_num_threads(1)
    from oracle import demo_anchor as da
    from oracle import gpflow05 as orc
    d = np.load(os.path.join(HERE, "init_liv_real_audio.npz"))
    t0 = time.time()
    r = da.run_demo(d["y"], int(d["fs"]), record_at=SNAP_AT,
                    progress=lambda it, f: print("iteration %5d  logged objective %.6f  (%.0f s)" % (it, f, time.time() - t0),
                                                 flush=True))
    P = da.PRINTED
    print("fun      %.10f   printed %.10f   relative difference %.2e" % (r["fun"], P["fun"], abs(r["fun"] / P["fun"] - 1)))
    print("x[:3]   ", r["x"][:3], " printed", P["x_head"])
    print("x[-3:]  ", r["x"][-3:], " printed", P["x_tail"])
    print("jac[:3] ", r["jac"][:3], " printed", P["jac_head"])
    print("jac[-3:]", r["jac"][-3:], " printed", P["jac_tail"])
    # constrained final state and the oracle's predictions at it (cell 11: xtest = x[::3])
    con = {k: (orc.positive_forward(r["free"][k]) if t == "+" else r["free"][k]) for k, t in da.ORDER}
    x, y, z, f0 = da.demo_inputs(d["y"], int(d["fs"]))
    kact = {"type": "matern32", "variance": float(con["act.variance"]), "lengthscales": float(con["act.lengthscales"]),
            "energy": [], "frequency": []}
    kcom = {"type": "mercer_matern12sm", "variance": float(con["com.variance"]),
            "lengthscales": float(con["com.lengthscales"]),
            "energy": [float(con["com.energy%d" % i]) for i in range(5)],
            "frequency": [float(con["com.frequency%d" % i]) for i in range(5)]}
    xt = x[::3].copy()
    ma, va, mc, vc, ms = orc.pdgp_predict_act_n_com(xt, [z[0][0]], [z[1][0]], [kact], [kcom], [con["q_mu_act"]],
                                                    [con["q_sqrt_act"]], [con["q_mu_com"]], [con["q_sqrt_com"]])
    out = dict(fun=r["fun"], x=r["x"], jac=r["jac"], logf=r["logf"], draws=r["draws"], idx_final=r["idx_final"],
               z=r["z"], snap_at=np.array(SNAP_AT), snaps=np.stack([r["snaps"][i] for i in SNAP_AT]),
               mean_a=ma[0], var_a=va[0], mean_c=mc[0], var_c=vc[0], mean_src=ms[0],
               printed_fun=P["fun"], printed_x_head=P["x_head"], printed_x_tail=P["x_tail"],
               printed_jac_head=P["jac_head"], printed_jac_tail=P["jac_tail"])
    for k, _ in da.ORDER:
        out["final." + k] = np.asarray(con[k])
    np.savez_compressed(os.path.join(HERE, "demo_real_audio_anchor.npz"), **out)
    print("wrote demo_real_audio_anchor.npz")


if __name__ == "__main__":
    main()
