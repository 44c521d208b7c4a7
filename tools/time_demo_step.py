"""Time the reference's own demo configuration (demos/scripts/demo-modgp.py: n = 16000 frames, minibatch_size = 100, inducing
points from init_liv, one pitch, 3 partials, Adam): steps per second of model.optimize().
    python tools/time_demo_step.py [--minibatch 100 --iters 1000]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minibatch", type=int, default=100)
    ap.add_argument("--iters", type=int, default=1000)
    args = ap.parse_args()
    import torch
    import gpitch_amd
    from gpitch_amd.kernels import Matern32
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    n, fs = 16000, 16000
    x = np.linspace(0., (n - 1.) / fs, n).reshape(-1, 1)
    f = sum(np.sin(2 * np.pi * x * (i + 1) * 15.) for i in range(3))
    component = f / np.max(np.abs(f))
    envelope = np.exp(-25 * (x - 0.33) ** 2) + np.exp(-75 * (x - 0.66) ** 2)
    envelope /= np.max(np.abs(envelope))
    y = component * envelope + np.sqrt(1e-6) * np.random.RandomState(0).randn(n, 1)
    z, u = gpitch_amd.init_liv(x=x, y=y, win_size=31, thres=0.05, dec=1)
    kact = Matern32(1, lengthscales=1.0, variance=1.0)
    kcom = MercerMatern12sm(1, energy=np.array([1., 1., 1.]), frequency=np.array([15., 30., 45.]))
    m = gpitch_amd.pdgp.Pdgp(x=x.copy(), y=y.copy(), z=z, kern=[[kact], [kcom]], minibatch_size=args.minibatch)
    m.za.fixed = True
    m.zc.fixed = True
    opt = gpitch_amd.train.AdamOptimizer(learning_rate=0.005)
    m.optimize(method=opt, maxiter=20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.optimize(method=opt, maxiter=args.iters)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("inducing points %d, minibatch %d: %d Adam steps in %.3f s = %.0f steps/s (%.3f ms per step)"
          % (z[0][0].shape[0] if isinstance(z, (list, tuple)) and isinstance(z[0], (list, tuple)) else np.asarray(z).shape[-2],
             args.minibatch, args.iters, dt, args.iters / dt, dt / args.iters * 1e3))


if __name__ == "__main__":
    main()
