"""Host side of an optimiser step: how long the Python / ctypes / HIP-launch work of one Adam step
takes against the device time of that step.  The host runs ahead of the device (nothing in a step synchronises), so the
device waits for the host only if the first number exceeds the second.
    python tools/host_profile.py [--M 256 --partials 5 --f32]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=32768)
    ap.add_argument("--M", type=int, default=256)
    ap.add_argument("--P", type=int, default=12)
    ap.add_argument("--partials", type=int, default=5)
    ap.add_argument("--f32", action="store_true")
    ap.add_argument("--steps", type=int, default=50)
    args = ap.parse_args()
    import torch
    import gpitch_amd
    from gpitch_amd.synth import make_problem, pdgp_from_problem
    model = pdgp_from_problem(make_problem(args.N, args.M, args.P, num_partials=args.partials, seed=0),
                              float_type=np.float32 if args.f32 else None)
    model.za.fixed = True; model.zc.fixed = True
    opt = gpitch_amd.train.AdamOptimizer(0.0025)
    model.optimize(method=opt, maxiter=5)          # warm-up, compiles the plan
    torch.cuda.synchronize()
    # pieces of the host side
    t = time.perf_counter()
    for _ in range(args.steps):
        idx = model.x.next_indices(); model.y.next_indices(); idx = np.sort(idx, kind="stable")
    t_idx = (time.perf_counter() - t) / args.steps
    torch.cuda.synchronize()
    import ctypes as C
    h = model._handle
    flag = C.c_int32(0)
    t0 = time.perf_counter()
    for _ in range(args.steps):                 # the body of Pdgp.optimize's Adam loop
        model._elbo(True, sync=False)
        model._adam_t += 1
        h.check(h.lib.gp_adam_step(h.h, model._free.data_ptr(), model._params.data_ptr(), model._grad.data_ptr(),
                                   model._tcode.data_ptr(), model._adam_m.data_ptr(), model._adam_v.data_ptr(),
                                   model._nparams, model._adam_t, opt.learning_rate, opt.beta1, opt.beta2, opt.epsilon))
        h.check(h.lib.gp_poll_not_pd(h.h, C.byref(flag)))
    t_host = (time.perf_counter() - t0) / args.steps      # everything enqueued; the device is still running
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / args.steps
    print("N=%d M=%d P=%d m=%d %s: %.3f ms per step on the device, %.3f ms of host work per step to enqueue it "
          "(index draw + sort %.3f ms of that)" % (args.N, args.M, args.P, args.partials, "f32" if args.f32 else "f64",
                                                    t_total * 1e3, t_host * 1e3, t_idx * 1e3))


if __name__ == "__main__":
    main()
