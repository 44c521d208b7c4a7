"""Window-fit throughput: SGPRSS fits of many small windows (ws = 2001 frames), the way
transcription.py:265-288 loops them, sequentially and with several HIP streams."""
import argparse, os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_windows(nwin, ws, M, P, m, seed=0):
    from gpitch_amd.synth import make_problem
    out = []
    for w in range(nwin):
        prob = make_problem(ws, M, P, num_partials=m, seed=seed + w)
        out.append((prob["x"], prob["y"], prob["zc"][0], prob))
    return out


def build_model(prob, handle):
    import gpitch_amd
    from gpitch_amd.kernels import Add
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.sgpr_ss import SGPRSS
    ks = [MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=1.0,
                           lengthscales=d["lengthscales"]) for d in prob["kern_com"]]
    return SGPRSS(prob["x"], prob["y"], Add(ks), prob["zc"][0], handle=handle)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nwin", type=int, default=16)
    ap.add_argument("--ws", type=int, default=2001)
    ap.add_argument("--M", type=int, default=64)
    ap.add_argument("--P", type=int, default=3)
    ap.add_argument("--m", type=int, default=10)
    ap.add_argument("--maxiter", type=int, default=10)
    ap.add_argument("--streams", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--batches", type=int, nargs="+", default=[32, 64, 128, 256], help="windows per batched evaluation")
    ap.add_argument("--nwin-batched", type=int, default=512)
    ap.add_argument("--json", default=None, help="write the measurements here")
    ap.add_argument("--predict", action="store_true", help="also time fit + predict_f + predict_s (SoSp's loop body)")
    ap.add_argument("--nwin-predict", type=int, default=256)
    args = ap.parse_args()
    import torch
    from gpitch_amd import _lib
    wins = make_windows(args.nwin, args.ws, args.M, args.P, args.m)
    import ctypes as C

    def time_eval(h, label):
        model = build_model(wins[0][3], h)
        model._compile(); model._pack()
        ps = model._param_list()
        x0 = np.array([p.transform.backward(p.value)[0] for p in ps if not p.fixed])
        for _ in range(3):
            model._objective(x0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            model._objective(x0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 100 * 1e3
        c = [C.c_int64() for _ in range(3)]
        h.lib.gp_sgpr_eval_counts(model._plan, *[C.byref(v) for v in c])
        print("one bound+grad evaluation, %s: %.3f ms  (eager %d / captured %d / replayed %d)"
              % (label, dt, c[0].value, c[1].value, c[2].value))
        model._destroy()

    time_eval(_lib.default_handle(), "null stream, eager launches")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hs = _lib.Handle(torch.cuda.current_device(), stream=s)
        time_eval(hs, "own stream, hipGraph replay")
        hs.close()

    out = {"config": {"ws": args.ws, "M": args.M, "kernels": args.P, "partials": args.m, "maxiter": args.maxiter},
           "streams": {}, "batched": {}}
    from gpitch_amd.windows import fit_windows, fit_windows_batched
    # device-batched fits (gp_sgprb_*): W windows per launch sequence, scipy's L-BFGS-B per window in reverse communication
    wins_b = make_windows(args.nwin_batched, args.ws, args.M, args.P, args.m, seed=1000)
    data_b = [(w[0], w[1], w[2]) for w in wins_b]
    for B in args.batches:
        fit_windows_batched(lambda hh: build_model(wins_b[0][3], hh), data_b[:B], maxiter=2, batch=B)   # warm-up
        t0 = time.perf_counter()
        res = fit_windows_batched(lambda hh: build_model(wins_b[0][3], hh), data_b, maxiter=args.maxiter, batch=B)
        dt = time.perf_counter() - t0
        nfev = sum(r["nfev"] for r in res)
        print("batched B=%d: %d windows in %.3f s = %.1f windows/s; nfev total %d (%.1f per window); bound[0]=%.6f"
              % (B, len(res), dt, len(res) / dt, nfev, nfev / float(len(res)), res[0]["bound"]), flush=True)
        out["batched"][str(B)] = {"windows": len(res), "seconds": dt, "windows_per_s": len(res) / dt, "nfev": nfev}
    # the whole loop body of SoSp.optimize (separation.py:279-313): fit, predict_f, predict_s for every window
    if args.predict:
        B = args.batches[-1]
        nwp = min(args.nwin_batched, args.nwin_predict)
        fit_windows_batched(lambda hh: build_model(wins_b[0][3], hh), data_b[:8], maxiter=2, batch=8, predict=True)   # warm-up
        t0 = time.perf_counter()
        res = fit_windows_batched(lambda hh: build_model(wins_b[0][3], hh), data_b[:nwp], maxiter=args.maxiter, batch=B,
                                  predict=True)
        dt = time.perf_counter() - t0
        print("batched fit + predict_f + predict_s, B=%d: %d windows in %.3f s = %.1f windows/s" % (B, nwp, dt, nwp / dt),
              flush=True)
        out["batched_with_predictions"] = {"batch": B, "windows": nwp, "seconds": dt, "windows_per_s": nwp / dt}
        # the same predictions from the one-window engine (what fit_windows' after_fit hook would call), 16 windows
        model = build_model(wins_b[0][3], _lib.default_handle())
        k = min(16, nwp)
        model.predict_f(wins_b[0][0]); model.predict_s(wins_b[0][0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for w in wins_b[:k]:
            model.X, model.Y, model.Z = w[0], w[1], w[2]
            model.predict_f(w[0]); model.predict_s(w[0])
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / k
        print("one-window engine: predict_f + predict_s %.2f ms per window" % (dt1 * 1e3), flush=True)
        out["one_window_predictions_ms"] = dt1 * 1e3
        model._destroy()
    for ns in args.streams:
        t0 = time.perf_counter()
        res = fit_windows(lambda hh: build_model(wins[0][3], hh), [(w[0], w[1], w[2]) for w in wins],
                          maxiter=args.maxiter, num_streams=ns)
        dt = time.perf_counter() - t0
        print("streams=%d: %d windows in %.3f s = %.1f windows/s; nfev total %d; bound[0]=%.6f"
              % (ns, args.nwin, dt, args.nwin / dt, sum(r["nfev"] for r in res), res[0]["bound"]), flush=True)
        out["streams"][str(ns)] = {"windows": args.nwin, "seconds": dt, "windows_per_s": args.nwin / dt}
    if args.json:
        import json
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
