"""child process of tests/test_gpu_sgpr.py::test_frame_sharded_optimize_with_a_trainable_mean_function_under_a_process_group:
one rank of a frame-sharded SGPRSS window with a trainable Linear mean function, optimised under torch.distributed
(nccl = RCCL with one rank, gloo with two ranks on the one GPU of the test box); rank 0 prints the result as JSON."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(shard):
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.mean_functions import Linear
    from gpitch_amd.sgpr_ss import SGPRSS
    from test_gpu_sgpr import _problem
    X, Y, Z, kl = _problem(1500, 32, 2, 5)
    Y = Y + 0.3 + 2.0 * (X - X.mean()) / np.ptp(X)              # something for the mean function to explain
    ks = [MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=d["variance"],
                           lengthscales=d["lengthscales"]) for d in kl]
    m = SGPRSS(X, Y, np.sum(ks), Z, mean_function=Linear(), shard=shard)
    m.likelihood.variance = 0.3
    return m


def main():
    backend = sys.argv[1]
    from gpitch_amd import dist as gd
    world, rank, _ = gd.env_world()
    import torch
    import torch.distributed as dist
    dist.init_process_group(backend)
    if torch.cuda.device_count() == 1:
        os.environ["LOCAL_RANK"] = "0"              # the rehearsal: every rank on the box's one GPU (the default handle's device)
    m = build((rank, world))
    res = m.optimize(maxiter=4)
    out = {"fun": float(res.fun), "x": [float(v) for v in res.x], "nfev": int(res.nfev), "world": world, "backend": backend}
    if rank == 0:
        ref = build(None)                                            # the unsharded window on the same GPU
        r0 = ref.optimize(maxiter=4)
        out["ref_fun"] = float(r0.fun)
        out["ref_x"] = [float(v) for v in r0.x]
        print("RESULT " + json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
