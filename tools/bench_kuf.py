"""Kuf strip build alone on the device (forward passes, helper stream off): average launch time of the spectral-mixture
and the stationary family for m partials.  Same-box A/B of build variants: GPITCH_AMD_LIB=tools/ab/lib_X.so python tools/bench_kuf.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from gpitch_amd.synth import make_problem, pdgp_from_problem
    N, M, P = 32768, 512, 12
    out = []
    for m, ft in ((20, None), (5, None), (20, np.float32)):
        model = pdgp_from_problem(make_problem(N, M, P, num_partials=m, seed=0), float_type=ft)
        model._pack()
        h = model._handle
        h.check(h.lib.gp_pdgp_set_overlap(model._plan, 0))
        for _ in range(3):
            model._elbo(False, sync=False)
        torch.cuda.synchronize()
        h.check(h.lib.gp_timers_enable(h.h, 1)); h.check(h.lib.gp_timers_reset(h.h))
        for _ in range(20):
            model._elbo(False, sync=False)
        torch.cuda.synchronize()
        h.check(h.lib.gp_timers_enable(h.h, 0))
        t = h.timers()
        T = 4 if ft is not None else 8
        for name, mm in (("kuf_build_sm", m), ("kuf_build", 0)):
            ms, n = t[name]
            byts = P * (T * (float(M) * N + N + M) + 8 * 2.0 * mm * (M + N))
            out.append("%s m=%d %s: %.4f ms = %.2f TB/s (%.3f of 8)" % (name, m, "f32" if ft else "f64", ms / n, byts / (ms / n * 1e-3) / 1e12,
                                                                        byts / (ms / n * 1e-3) / 8e12))
        del model
        torch.cuda.empty_cache()
    print("lib=%s | " % os.environ.get("GPITCH_AMD_LIB", "product") + " | ".join(out))


if __name__ == "__main__":
    main()
