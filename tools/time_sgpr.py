"""Time one SGPRSS bound + gradient evaluation (BASELINE configs[4]: 5 sources, N = 65536, M = 512) in float64 / float32,
with the library's per-class kernel timers.
    python tools/time_sgpr.py [--N 65536 --M 512 --P 5 --m 3]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=65536)
    ap.add_argument("--M", type=int, default=512)
    ap.add_argument("--P", type=int, default=5)
    ap.add_argument("--m", type=int, default=3)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--only", choices=("f64", "f32"), default=None)
    args = ap.parse_args()
    import torch
    from gpitch_amd import _lib
    from gpitch_amd.kernels import Add
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.sgpr_ss import SGPRSS
    rng = np.random.RandomState(0)
    X = np.linspace(0, (args.N - 1) / 16000., args.N).reshape(-1, 1)
    Y = rng.randn(args.N, 1)
    Z = X[:: args.N // args.M][:args.M].copy()
    h = _lib.default_handle()
    for ft in (np.float64, np.float32):
        if args.only and ft.__name__ != "float" + args.only[1:]:
            continue
        ks = [MercerMatern12sm(1, energy=np.full(args.m, 1.0 / args.m), frequency=110.0 * (p + 1) * np.arange(1, args.m + 1),
                               variance=1.0, lengthscales=0.05 + 0.01 * p) for p in range(args.P)]
        model = SGPRSS(X, Y, Add(ks), Z, handle=h, float_type=ft)
        model._compile(); model._pack()
        g = h.zeros(model._nparams)
        for _ in range(3):
            model._bound(grad=g)
        torch.cuda.synchronize()
        h.check(h.lib.gp_timers_enable(h.h, 1)); h.check(h.lib.gp_timers_reset(h.h))
        t0 = time.perf_counter()
        for _ in range(args.reps):
            f = model._bound(grad=g)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.reps * 1e3
        h.check(h.lib.gp_timers_enable(h.h, 0))
        tm = {k: round(ms / args.reps, 3) for k, (ms, n) in h.timers().items() if n}
        print("%s: %.3f ms per bound+gradient evaluation, bound %.6f; kernel classes (ms): %s" % (ft.__name__, dt, f, tm), flush=True)
        model._destroy()


if __name__ == "__main__":
    main()
