"""bench.py — ELBO-steps/sec of the Pdgp hot path on MI355X (BASELINE.json metric).

One step = what one iteration of gpflow Model.optimize(method=Adam) does in the reference
(SURVEY §3.2): batch -> forward ELBO -> gradient w.r.t. every non-fixed parameter -> Adam update of
the free state.  Workload: N=32768 frames, M=512 inducing points per latent GP, P=12 pitches (24
latent GPs), float64, inputs resident in HBM.  The batch is the FULL data set (minibatch_size = N):
the reference's MinibatchData would hand over a permutation of it every step; the ELBO is a sum over
the batch, so the engine takes the resident x, y in time order — no index draw, sort or gather inside
the timed step (gpitch_amd/pdgp.py:_batch; equality with the permuted batch: tests/test_gpu_fullsize.py).

Multi-GPU (torchrun, one rank per GPU): the reference scales N by independent windows / segments
(window_overlap.py:194, transcription.py:265-288), so every rank owns one independent 32768-frame
window with the full 12-pitch model (weak scaling) and the scalar ELBO is all-reduced over RCCL each
step; there is no other data-path collective.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MFMA_TFLOPS = 78.6   # MI355X FP64 matrix peak (vendor sheet); a bare VGPR-accumulator MFMA loop measures 70-76
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X FP32 matrix peak (v_mfma_f32_16x16x4_f32: 64 FLOP/clk/SIMD; guide: 155 measured)
PEAK_HBM_GBS = 8000.0         # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def _switch(name, default):
    """value of one library switch (GPITCH_AMD_SWITCHES="name=value,...": gpitch_amd/csrc/switches.h)"""
    for item in os.environ.get("GPITCH_AMD_SWITCHES", "").split(","):
        if item.startswith(name + "="):
            try:
                return int(item.split("=", 1)[1])
            except ValueError:
                pass
    return default


def build_model(args, rank):
    import gpitch_amd
    from gpitch_amd.pdgp import Pdgp
    from gpitch_amd.synth import make_problem, pdgp_from_problem
    ft = {"f32": np.float32, "mixed": (np.float64, np.float32)}.get(args.float_type, np.float64)
    if args.shard in ("pitch", "gp"):
        # ONE model over all ranks: same problem everywhere; rank r holds pitches p = r (mod world) ("pitch": both GPs of a
        # pitch, all-reduce of 3N+1) or latent GPs g = r (mod world) ("gp": all-gather of (fmean, fvar) per GP)
        world = int(os.environ.get("WORLD_SIZE", "1"))
        prob = make_problem(args.N, args.M, args.P, num_partials=args.partials, seed=0)
        import torch.distributed as tdist
        grouped = world > 1 or (tdist.is_available() and tdist.is_initialized())     # (a one-rank group: the RCCL rehearsal)
        sh = ((rank, world) if args.shard == "pitch" else ("gp", rank, world)) if grouped else None
        model = pdgp_from_problem(prob, shard=sh, float_type=ft)
    else:
        prob = make_problem(args.N, args.M, args.P, num_partials=args.partials, seed=rank)
        model = pdgp_from_problem(prob, float_type=ft)
    model.za.fixed = True      # as demos/scripts/demo-modgp.py:40-41
    model.zc.fixed = True
    return prob, model


def cpu_baseline(args):
    """The oracle's torch-CPU float64 restatement of the GPflow graph (autograd backward + Adam), timed on this
    host's cores by the SURVEY section 8d protocol: 2 warm-up steps, median of >= 5 timed steps (min / max kept).
    Bounded sample: one pitch (2 of the 2P latent GPs) at the full N and M; a step over P pitches costs P times
    that (the 2P conditionals are independent and carry > 95 %).  --cpu-full also times whole P-pitch steps so the
    1/P scaling can be checked (about a minute per step: off by default, result committed under profiles/)."""
    import torch
    from gpitch_amd.synth import make_problem
    from oracle import gpflow05 as orc
    from oracle.backend import TorchBackend
    tb = TorchBackend()

    def make_step(P):
        prob = make_problem(args.N, args.M, P, num_partials=args.partials, seed=0)
        T = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)
        x, y = torch.tensor(prob["x"]), torch.tensor(prob["y"])
        leaves = []

        def tk(d):
            out = dict(d)
            for key in ("variance", "lengthscales"):
                out[key] = T(d[key]); leaves.append(out[key])
            out["energy"] = [T(e) for e in d["energy"]]; leaves.extend(out["energy"])
            out["frequency"] = [T(f) for f in d["frequency"]]; leaves.extend(out["frequency"])
            return out
        ka, kc = [tk(d) for d in prob["kern_act"]], [tk(d) for d in prob["kern_com"]]
        qma, qmc = [T(q) for q in prob["q_mu_act"]], [T(q) for q in prob["q_mu_com"]]
        qsa, qsc = [T(q) for q in prob["q_sqrt_act"]], [T(q) for q in prob["q_sqrt_com"]]
        nv = T(prob["noise_var"])
        leaves += qma + qmc + qsa + qsc + [nv]
        za, zc = [torch.tensor(z) for z in prob["za"]], [torch.tensor(z) for z in prob["zc"]]
        mom = [(torch.zeros_like(l), torch.zeros_like(l)) for l in leaves]

        def step(t):
            for l in leaves:
                l.grad = None
            elbo = orc.pdgp_elbo(x, y, za, zc, ka, kc, qma, qsa, qmc, qsc, nv, whiten=True, xp=tb)
            (-elbo).backward()
            with torch.no_grad():
                lr_t = 0.0025 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
                for l, (m, v) in zip(leaves, mom):
                    m.mul_(0.9).add_(l.grad, alpha=0.1)
                    v.mul_(0.999).addcmul_(l.grad, l.grad, value=0.001)
                    l.sub_(lr_t * m / (v.sqrt() + 1e-8) * 1e-3)   # tiny steps: stay in the positive region
        return step

    def timed(step, warm, n):
        for t in range(1, 1 + warm):
            step(t)
        out = []
        for t in range(1 + warm, 1 + warm + n):
            t0 = time.perf_counter()
            step(t)
            out.append(time.perf_counter() - t0)
        return out
    times = timed(make_step(1), 2, max(args.cpu_steps, 1))
    t1 = float(np.median(times))
    threads = int(torch.get_num_threads())
    res = {"value": 1.0 / (t1 * args.P), "unit": "ELBO-steps/sec", "cores": threads,
           "kind": "port",
           "sample": "2 warm-ups then median of %d steps of the P=1 sub-problem (2 of %d latent GPs) at N=%d, M=%d, m=%d, "
                     "torch-CPU float64 + autograd + Adam (median %.2f s, min %.2f, max %.2f); scaled by 1/P for the "
                     "P=%d step; torch intra-op threads=%d (torch's default: one per physical core; the host "
                     "reports %d logical CPUs, SMT siblings add nothing to float64 GEMM)"
                     % (len(times), 2 * args.P, args.N, args.M, args.partials, t1, min(times), max(times), args.P,
                        threads, os.cpu_count()),
           "step_seconds_p1": {"median": t1, "min": float(min(times)), "max": float(max(times)), "n": len(times)}}
    if args.cpu_full > 0:
        full = timed(make_step(args.P), 1, args.cpu_full)
        tf = float(np.median(full))
        res["full_step_check"] = {"P": args.P, "seconds": full, "median": tf, "steps_per_sec": 1.0 / tf,
                                  "ratio_to_P_times_p1": tf / (t1 * args.P)}
    return res


def bench_cfg5(args, with_cpu):
    """BASELINE configs[4]: sgpr_ss source separation, 5 sources, N = 65536 frames, M = 512 inducing points — one evaluation =
    SGPRSS.build_likelihood (sgpr_ss.py:29-71) + its gradient w.r.t. every kernel hyper-parameter and the noise variance
    (what one L-BFGS-B function evaluation of model.optimize costs), float64 and float32 strips.  Algorithmic flops per
    evaluation: forward A = L^-1 Kuf (M^2 N) + A A^T (M^2 N, symmetric); backward twice that: 6 M^2 N."""
    import torch
    from gpitch_amd import _lib
    from gpitch_amd.kernels import Add
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.sgpr_ss import SGPRSS
    N, M, P, m = 65536, 512, 5, 3
    rng = np.random.RandomState(0)
    X = np.linspace(0, (N - 1) / 16000., N).reshape(-1, 1)
    Y = rng.randn(N, 1)
    Z = X[:: N // M][:M].copy()
    # a stream of its own: gp_sgpr_bound_grad then replays its ~60 dependent launches from a hipGraph (the legacy null
    # stream cannot be captured)
    dev0 = _lib.default_handle().device
    stream = torch.cuda.Stream(device=dev0)
    torch.cuda.set_stream(stream)
    h = _lib.Handle(dev0.index, stream=stream)
    flops = 6.0 * M * M * N
    kuf_bytes = P * 8.0 * (float(M) * N + N + M) + P * 8.0 * 2 * m * (M + N)
    out = {"workload": "sgpr_ss bound + gradient evaluation, N=%d x M=%d, %d Mercer Matern-1/2 SM kernels (m=%d) "
                       "(BASELINE configs[4])" % (N, M, P, m),
           "unit": "evaluations/s", "algorithmic_flops_per_evaluation": flops,
           "kuf_build_bytes_per_evaluation_f64": kuf_bytes}

    def kernels():
        return [MercerMatern12sm(1, energy=np.full(m, 1.0 / m), frequency=110.0 * (p + 1) * np.arange(1, m + 1),
                                 variance=1.0, lengthscales=0.05 + 0.01 * p) for p in range(P)]
    for ft, name, peak in ((np.float64, "f64", PEAK_F64_MFMA_TFLOPS), (np.float32, "f32", PEAK_F32_MFMA_TFLOPS)):
        model = SGPRSS(X, Y, Add(kernels()), Z, handle=h, float_type=ft)
        model._compile(); model._pack()
        g = h.zeros(model._nparams)
        for _ in range(4):                     # eager, capture, replay, replay
            model._bound(grad=g)
        torch.cuda.synchronize()
        reps = max(args.steps, 5)
        t0 = time.perf_counter()
        for _ in range(reps):
            f = model._bound(grad=g)           # (returns the bound: one host synchronisation per evaluation, as L-BFGS-B needs)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        # kernel classes from the library's event timers (eager launches: timers switch the graph replay off)
        h.check(h.lib.gp_timers_enable(h.h, 1)); h.check(h.lib.gp_timers_reset(h.h))
        t0 = time.perf_counter()
        for _ in range(reps):
            model._bound(grad=g)
        torch.cuda.synchronize()
        dt_eager = (time.perf_counter() - t0) / reps
        h.check(h.lib.gp_timers_enable(h.h, 0))
        tm = {k: ms / reps for k, (ms, n) in h.timers().items() if n}
        out[name] = {"value": 1.0 / dt, "ms_per_evaluation": dt * 1e3, "ms_per_evaluation_eager_launches": dt_eager * 1e3,
                     "bound": float(f),
                     "roofline": {"bound": "mfma", "achieved": flops / dt / 1e12, "peak": peak, "unit": "TFLOP/s",
                                  "frac": flops / dt / 1e12 / peak,
                                  "note": "whole evaluation, replayed from a hipGraph (its strip products are a quarter of it: "
                                          "the rest is the chain of dependent M x M launches)"},
                     "kernel_ms_per_evaluation": tm}
        model._destroy()
    stream.synchronize()
    torch.cuda.set_stream(torch.cuda.default_stream(dev0))
    h.close()
    if with_cpu:
        from oracle import gpflow05 as orc
        from oracle.backend import TorchBackend
        tb = TorchBackend()
        T = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)
        kl = [{"type": "mercer_matern12sm", "variance": T(1.0), "lengthscales": T(0.05 + 0.01 * p),
               "energy": [T(1.0 / m) for _ in range(m)], "frequency": [T(110.0 * (p + 1) * (q + 1)) for q in range(m)]}
              for p in range(P)]
        nv = T(1.0)
        Xt, Yt, Zt = torch.tensor(X), torch.tensor(Y), torch.tensor(Z)
        times = []
        for it in range(4):
            t0 = time.perf_counter()
            b = orc.sgpr_bound(Xt, Yt, Zt, kl, nv, xp=tb)
            b.backward()
            times.append(time.perf_counter() - t0)
        t1 = float(np.median(times[1:]))
        out["cpu_baseline"] = {"value": 1.0 / t1, "unit": "evaluations/s", "cores": int(torch.get_num_threads()), "kind": "port",
                               "sample": "1 warm-up then median of 3 bound + autograd-gradient evaluations of the oracle's "
                                         "torch-CPU float64 restatement at the full size (%.2f s each)" % t1}
        out["vs_cpu_baseline"] = out["f64"]["value"] / out["cpu_baseline"]["value"]
    return out


def bench_windows(args, with_cpu):
    """The window loop of AMT.optimize / SoSp.optimize (transcription.py:265-288): independent SGPRSS fits of ws = 2001-frame
    windows (M = 64 inducing points, 3 pitch kernels of 10 partials, L-BFGS-B maxiter 10), device-batched 256 windows per
    launch sequence (gp_sgprb_*), scipy's L-BFGS-B per window."""
    import torch
    from gpitch_amd.kernels import Add
    from gpitch_amd.matern12_spectral_mixture import MercerMatern12sm
    from gpitch_amd.sgpr_ss import SGPRSS
    from gpitch_amd.synth import make_problem
    from gpitch_amd.windows import fit_windows_batched
    ws, M, P, m, maxiter, nwin, B = 2001, 64, 3, 10, 10, 512, 256
    probs = [make_problem(ws, M, P, num_partials=m, seed=1000 + w) for w in range(nwin)]
    data = [(q["x"], q["y"], q["zc"][0]) for q in probs]

    def make(hh):
        ks = [MercerMatern12sm(1, energy=np.array(d["energy"]), frequency=np.array(d["frequency"]), variance=1.0,
                               lengthscales=d["lengthscales"]) for d in probs[0]["kern_com"]]
        return SGPRSS(probs[0]["x"], probs[0]["y"], Add(ks), probs[0]["zc"][0], handle=hh)
    fit_windows_batched(make, data[:B], maxiter=2, batch=B)        # warm-up (plan, graph capture)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = fit_windows_batched(make, data, maxiter=maxiter, batch=B)
    dt = time.perf_counter() - t0
    nfev = sum(r["nfev"] for r in res)
    out = {"workload": "sgpr_ss window fits (transcription.py:265-288): ws=%d frames, M=%d, %d kernels x %d partials, L-BFGS-B "
                       "maxiter %d, %d windows, %d per launch sequence" % (ws, M, P, m, maxiter, nwin, B),
           "value": nwin / dt, "unit": "window fits/s", "seconds": dt, "evaluations_per_window": nfev / float(nwin),
           "failed_windows": sum(1 for r in res if "error" in r)}
    if with_cpu:
        from oracle import gpflow05 as orc
        from oracle.backend import TorchBackend
        tb = TorchBackend()
        T = lambda a: torch.tensor(np.asarray(a, dtype=np.float64), requires_grad=True)
        q = probs[0]
        kl = [{"type": "mercer_matern12sm", "variance": T(1.0), "lengthscales": T(d["lengthscales"]),
               "energy": [T(e) for e in d["energy"]], "frequency": [T(f) for f in d["frequency"]]} for d in q["kern_com"]]
        nv = T(1.0)
        Xt, Yt, Zt = torch.tensor(q["x"]), torch.tensor(q["y"]), torch.tensor(q["zc"][0])
        times = []
        for it in range(12):
            t0 = time.perf_counter()
            b = orc.sgpr_bound(Xt, Yt, Zt, kl, nv, xp=tb)
            b.backward()
            times.append(time.perf_counter() - t0)
        te = float(np.median(times[2:]))
        per_win = te * out["evaluations_per_window"]
        out["cpu_baseline"] = {"value": 1.0 / per_win, "unit": "window fits/s", "cores": int(torch.get_num_threads()),
                               "kind": "port",
                               "sample": "median of 10 bound + autograd-gradient evaluations of one window by the oracle's "
                                         "torch-CPU restatement (%.1f ms each) x the %.1f evaluations a fit takes"
                                         % (te * 1e3, out["evaluations_per_window"])}
        out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    return out


def self_launch(args):
    """`python bench.py --gpus N` with no launcher around it: start N ranks (one per GPU) as fresh child processes
    through torch.distributed.run BEFORE this process makes any HIP call (a process that has touched the GPU must
    never be re-exec'd on this pool), relay their output (rank 0 prints the JSON line) and return their exit code."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.call(cmd, env=env)


def run_timed(args, shard, rank, dist):
    """warm-up, then exactly args.steps steps bracketed by barrier + synchronize on both sides; MAX over ranks."""
    import torch
    import gpitch_amd
    from gpitch_amd import dist as gp_dist
    sargs = argparse.Namespace(**vars(args))
    sargs.shard = shard
    prob, model = build_model(sargs, rank)
    model._pack()
    h = model._handle
    h.check(h.lib.gp_pdgp_set_overlap(model._plan, args.overlap))
    opt = gpitch_amd.train.AdamOptimizer(args.lr)
    elbo_sum = torch.zeros(1, dtype=torch.float64, device=h.device)

    def step():
        model._elbo(True, sync=False)
        model._adam_t += 1
        h.check(h.lib.gp_adam_step(h.h, model._free.data_ptr(), model._params.data_ptr(), model._grad.data_ptr(),
                                   model._tcode.data_ptr(), model._adam_m.data_ptr(), model._adam_v.data_ptr(),
                                   model._nparams, model._adam_t, opt.learning_rate, opt.beta1, opt.beta2, opt.epsilon))
        if dist is not None and shard == "window":
            # scalar ELBO of the whole job (north_star: all-reduce of the scalar ELBO)
            elbo_sum.copy_(model._elbo_dev[:1])
            gp_dist.allreduce_sum_(elbo_sum)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    h.check(h.lib.gp_timers_enable(h.h, 1))
    h.check(h.lib.gp_timers_reset(h.h))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    h.check(h.lib.gp_timers_enable(h.h, 0))
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=h.device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
    return {"model": model, "handle": h, "elapsed": elapsed, "elbo_final": float(model._elbo_dev[0].item()),
            "timers": h.timers()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--N", type=int, default=32768)
    ap.add_argument("--M", type=int, default=512)
    ap.add_argument("--P", type=int, default=12)
    ap.add_argument("--partials", type=int, default=20)
    ap.add_argument("--lr", type=float, default=0.0025)
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed CPU-baseline steps (after 2 warm-ups)")
    ap.add_argument("--cpu-full", type=int, default=1,
                    help="also time this many whole P-pitch CPU steps (validates the 1/P scaling; ~1 min each)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--shard", choices=["window", "pitch", "gp"], default="window",
                    help="window: one independent P-pitch window per GPU (weak scaling, scalar all-reduce only); "
                         "pitch: ONE P-pitch model spread over the GPUs, one all-reduce of 3N+1 doubles per step "
                         "(strong scaling, ceiling P / ceil(P / gpus)); gp: ONE model with its 2P latent GPs dealt over the "
                         "GPUs, one all-gather of (fmean, fvar) per GP per step (strong scaling, ceiling 2P / ceil(2P / gpus))")
    ap.add_argument("--overlap", type=int, choices=[0, 1, 2], default=2,
                    help="gp_pdgp_set_overlap level: 0 one stream (clean single-kernel timings), 1 Kuu-side work on the "
                         "helper stream, 2 (library default) also H = A D A^T next to Kuf_bar")
    ap.add_argument("--float-type", choices=["f64", "f32", "mixed"], default="f64",
                    help="f64 (the headline: the reference's float_type = float64); f32: the M x N strips and the four "
                         "strip products in float32 (BASELINE configs 3 and 5 are quoted at fp32; tolerance in tests/test_gpu_f32.py); "
                         "mixed: activation GPs float64, component GPs float32")
    ap.add_argument("--no-f32-line", action="store_true", help="skip the extra cfg3 (M=256, fp32) measurement")
    ap.add_argument("--no-sgpr-lines", action="store_true",
                    help="skip the extra cfg5 (sgpr_ss N=65536, M=512, 5 sources) and window-loop measurements")
    ap.add_argument("--no-pitch-line", action="store_true",
                    help="multi-GPU window mode: skip the extra pitch-sharded (strong-scaling) measurement")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(args.gpus, 1):
        sys.exit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one process per GPU; a rehearsal with more ranks than GPUs (gloo) wraps around
    dev_index = local_rank % max(torch.cuda.device_count(), 1)
    os.environ["LOCAL_RANK"] = str(dev_index)      # gpitch_amd's default handle binds to this device
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1 or ("MASTER_ADDR" in os.environ and "WORLD_SIZE" in os.environ):
        # (a launcher started us: the process group is formed even for one rank, which is how a one-GPU box exercises
        # RCCL — communicator set-up and the per-step all-reduce on the library's streams)
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)

    res = run_timed(args, args.shard, rank, dist)
    model, h, elapsed, elbo_final, timers = res["model"], res["handle"], res["elapsed"], res["elbo_final"], res["timers"]
    local_gps = model._gps()         # latent GPs in this rank's launches
    n_act_local = sum(1 for g in local_gps if any(g[0] is k for k in model.kern_act))
    n_com_local = len(local_gps) - n_act_local
    extra_pitch = extra_gp = None
    if world > 1 and args.shard == "window" and not args.no_pitch_line:
        # the same job as ONE model spread over the ranks (strong scaling): extra keys of the same JSON line
        model = res = None
        torch.cuda.empty_cache()
        rp = run_timed(args, "pitch", rank, dist)
        extra_pitch = {"value": args.steps / rp["elapsed"], "unit": "steps/s", "ms_per_step": rp["elapsed"] / args.steps * 1e3,
                       "scaling": "strong", "exchange": "one all-reduce of 3N+1 doubles per step",
                       "pitches_per_rank_max": -(-args.P // world), "ceiling_x": args.P / float(-(-args.P // world)),
                       "elbo_final": rp["elbo_final"]}
        del rp
        torch.cuda.empty_cache()
        rg = run_timed(args, "gp", rank, dist)
        extra_gp = {"value": args.steps / rg["elapsed"], "unit": "steps/s", "ms_per_step": rg["elapsed"] / args.steps * 1e3,
                    "scaling": "strong", "exchange": "one all-gather of 2N doubles per latent GP (+ the KL scalar) per step",
                    "latent_gps_per_rank_max": -(-2 * args.P // world),
                    "ceiling_x": 2 * args.P / float(-(-2 * args.P // world)), "elbo_final": rg["elbo_final"]}
        del rg

    forward_only = None
    if world == 1:
        # forward-only ELBO evaluations per second as a secondary line (fresh batch + ELBO, no gradient, no Adam)
        for _ in range(2):
            model._elbo(False, sync=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model._elbo(False, sync=False)
        torch.cuda.synchronize()
        dtf = time.perf_counter() - t0
        forward_only = {"value": args.steps / dtf, "unit": "ELBO evaluations/s", "ms_per_evaluation": dtf / args.steps * 1e3}
    kuf_alone = None
    if world == 1 and args.overlap != 0 and not args.no_f32_line:
        # the Kuf builds alone on the device (helper stream off), forward passes of the model just timed: in the step the
        # resident Kuu factorisation holds 24 of the 256 CUs beside them
        h.check(h.lib.gp_pdgp_set_overlap(model._plan, 0))
        model._elbo(False, sync=False)
        torch.cuda.synchronize()
        h.check(h.lib.gp_timers_enable(h.h, 1))
        h.check(h.lib.gp_timers_reset(h.h))
        for _ in range(5):
            model._elbo(False, sync=False)
        torch.cuda.synchronize()
        h.check(h.lib.gp_timers_enable(h.h, 0))
        kuf_alone = {k: h.timers()[k] for k in ("kuf_build", "kuf_build_sm")}
        h.check(h.lib.gp_pdgp_set_overlap(model._plan, args.overlap))
    kuf_m5 = None
    if world == 1 and (args.N, args.M, args.P) == (32768, 512, 12) and args.partials != 5 and not args.no_f32_line:
        # the HBM-bound regime of the spectral-mixture Kuf build (m = 5 partials, the demo's size): forward passes only
        model = res = None
        torch.cuda.empty_cache()
        from gpitch_amd.synth import make_problem, pdgp_from_problem
        m5 = pdgp_from_problem(make_problem(args.N, args.M, args.P, num_partials=5, seed=0))
        m5._pack()
        h5 = m5._handle
        for _ in range(2):
            m5._elbo(False, sync=False)
        torch.cuda.synchronize()
        h5.check(h5.lib.gp_timers_enable(h5.h, 1))
        h5.check(h5.lib.gp_timers_reset(h5.h))
        for _ in range(5):
            m5._elbo(False, sync=False)
        torch.cuda.synchronize()
        h5.check(h5.lib.gp_timers_enable(h5.h, 0))
        ms5, n5 = h5.timers()["kuf_build_sm"]
        # ... and with the helper stream off: the build alone on the device (in the step the resident Kuu factorisation
        # holds 24 of the 256 CUs beside it)
        h5.check(h5.lib.gp_pdgp_set_overlap(m5._plan, 0))
        m5._elbo(False, sync=False)
        torch.cuda.synchronize()
        h5.check(h5.lib.gp_timers_enable(h5.h, 1))
        h5.check(h5.lib.gp_timers_reset(h5.h))
        for _ in range(5):
            m5._elbo(False, sync=False)
        torch.cuda.synchronize()
        h5.check(h5.lib.gp_timers_enable(h5.h, 0))
        ms5a, n5a = h5.timers()["kuf_build_sm"]
        if n5:
            b5 = args.P * (8 * (float(args.M) * args.N + args.N + args.M) + 8 * 2.0 * 5 * (args.M + args.N))
            kuf_m5 = {"bound": "hbm", "partials": 5, "achieved": b5 / (ms5 / n5 * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                      "unit": "GB/s", "frac": b5 / (ms5 / n5 * 1e-3) / 1e9 / PEAK_HBM_GBS, "avg_launch_ms": ms5 / n5,
                      "algorithmic_bytes_per_launch": b5, "latent_gps_per_launch": args.P,
                      "note": "forward-only evaluations of the same model with m = 5 partials; in the step "
                              "(overlap level 2) the resident Kuu factorisation runs beside the build"}
            if n5a:
                kuf_m5["alone"] = {"avg_launch_ms": ms5a / n5a, "achieved": b5 / (ms5a / n5a * 1e-3) / 1e9,
                                   "frac": b5 / (ms5a / n5a * 1e-3) / 1e9 / PEAK_HBM_GBS, "overlap_level": 0}
        m5 = None
        torch.cuda.empty_cache()
    cfg3 = None
    if world == 1 and not args.no_f32_line and (args.N, args.M, args.P) == (32768, 512, 12) and args.float_type == "f64":
        # BASELINE configs[2] at its own precision: 12-pitch transcription model, N=32768, M=256 per pitch, fp32 (an extra
        # key; the headline above stays float64).  Same step, same timing protocol; float64 beside it.
        model = res = None
        torch.cuda.empty_cache()
        cfg3 = {"workload": "pdgp ELBO step, N=32768 x M=256 x P=12, m=5 partials (BASELINE configs[2])",
                "tolerance": "ELBO within 2e-4 relative of the float64 oracle (tests/test_gpu_f32.py: 4.7e-5 measured); "
                             "'mixed' = activation GPs float64, component GPs float32 (gp_pdgp_set_gp_precision): ELBO within 1e-7, "
                             "activation posterior means 2e-7, activation-side gradients 2e-6 (test_mixed_precision_*)"}
        for ft in ("f32", "mixed", "f64"):
            a3 = argparse.Namespace(**vars(args))
            a3.M, a3.partials, a3.float_type = 256, 5, ft
            r3 = run_timed(a3, "window", rank, None)
            tm = r3["timers"]
            gem = sum(tm[k][0] for k in ("cond_A", "cond_LTA", "nt_gemm", "kuf_bar")) / args.steps
            cfg3[ft] = {"value": args.steps / r3["elapsed"], "unit": "steps/s", "ms_per_step": r3["elapsed"] / args.steps * 1e3,
                        "strip_gemm_ms_per_step": gem, "elbo_final": r3["elbo_final"]}
            if ft != "mixed":       # whole step against the matrix peak of its strips' type (5 M^2 N per latent GP)
                pk = PEAK_F32_MFMA_TFLOPS if ft == "f32" else PEAK_F64_MFMA_TFLOPS
                tf = 5.0 * 256.0 * 256.0 * args.N * 2 * args.P / (r3["elapsed"] / args.steps) / 1e12
                cfg3[ft]["mfma_frac_step"] = {"achieved": tf, "peak": pk, "unit": "TFLOP/s", "frac": tf / pk}
            r3 = None
            torch.cuda.empty_cache()
        cfg3["dtype"] = "f32"
        cfg3["speedup_over_f64"] = cfg3["f32"]["value"] / cfg3["f64"]["value"]
    if rank == 0:
        pitch = args.shard in ("pitch", "gp") and world > 1          # ONE model over the ranks (strong scaling)
        G, M, N, T = len(local_gps), args.M, args.N, 8      # latent GPs in this rank's launches
        m2n = float(M) * M * N
        # algorithmic flops per step (SURVEY §8d / DESIGN.md) and per launch: a product is one launch over all 2P latent
        # GPs, except Kuf_bar, which is issued per kernel family (two launches of 12 GPs each at the default overlap
        # level) -> per-launch figures are averages over a step's launches of that kernel
        alg_step = {"cond_A": G * m2n, "cond_LTA": G * m2n, "nt_gemm": G * m2n, "kuf_bar": G * 2.0 * m2n}
        launches_per_step = {k: timers[k][1] / float(args.steps) for k in alg_step}
        alg = {k: alg_step[k] / max(launches_per_step[k], 1.0) for k in alg_step}
        per_launch = {k: (ms / max(n, 1)) for k, (ms, n) in timers.items()}
        dom = max(alg, key=lambda k: timers[k][0])
        dom_ms = per_launch[dom]
        f32 = args.float_type == "f32"
        PEAK = PEAK_F32_MFMA_TFLOPS if f32 else PEAK_F64_MFMA_TFLOPS
        T = 4 if f32 else 8
        if f32:
            sym = {"cond_A": "gemm_f32_kernel<1>", "cond_LTA": "gemm_f32_kernel<2>", "nt_gemm": "gemm_f32_kernel<4>",
                   "kuf_bar": "gemm_f32_kernel<3>"}
        else:
            # (whole aligned strips run gemm_strip.hip's lean forms; ragged shapes fall back to gemm_f64_kernel<128,128,...>)
            lean = (M % 128 == 0) and (N % 128 == 0) and _switch("strip_lean", 1) != 0
            # (... and, M a multiple of 64 and N of 256, gemm_wave.hip's: a 64 x 64 tile per wavefront, no LDS; the split-K
            #  product stays with gemm_strip_nt_kernel)
            wave = (M % 64 == 0) and (N % 256 == 0) and _switch("strip_wave", 1) != 0
            # (Kuf_bar is one launch per kernel family: the stationary activation family's — inducing inputs fixed —
            #  contracts its tile with dK/dtheta in the epilogue and stores nothing, gemm_strip_kernel<5, KT = Matern32>)
            fused = _switch("hyper_fuse", 1) != 0 and args.overlap >= 0
            sym = ({"cond_A": "gemm_strip_kernel<1,-1>", "cond_LTA": "gemm_strip_kernel<2,-1>", "nt_gemm": "gemm_strip_nt_kernel",
                    "kuf_bar": "gemm_strip_kernel<3,-1> + gemm_strip_kernel<5,1>" if fused else "gemm_strip_kernel<3,-1>"} if lean else
                   {"cond_A": "gemm_f64_kernel<128,128,false,false,1>", "cond_LTA": "gemm_f64_kernel<128,128,true,false,2>",
                    "nt_gemm": "gemm_f64_kernel<128,128,false,true,4>", "kuf_bar": "gemm_f64_kernel<128,128,false,false,3>"})
            if wave:
                sym.update({"cond_A": "gemm_wave_kernel<1,-1>", "cond_LTA": "gemm_wave_kernel<2,-1>",
                            "kuf_bar": "gemm_wave_kernel<3,-1> + gemm_wave_kernel<5,1>" if fused else "gemm_wave_kernel<3,-1>"})
        # roofline (by its definition): the dominant kernel's OWN algorithmic flops per launch / its OWN mean launch
        # duration (HIP events on the stream it is launched on).  At overlap level 2 other kernels share the chip with
        # it (the split-K product on the helper stream), which lengthens its launch: that shows up here, undisguised.
        achieved = alg[dom] / (dom_ms * 1e-3) / 1e12 if dom_ms > 0 else 0.0
        traffic, traffic_src = None, None
        try:   # HBM bytes per launch from the committed rocprofv3 --pmc passes of this command (tools/make_traffic_json.py)
            tfile = os.path.join("profiles", "r04", "hbm_traffic.json")
            tj = json.load(open(os.path.join(ROOT, tfile)))
            if (N, M, G, args.partials) == (32768, 512, 24, 20) and tj.get("overlap_level", 2) == args.overlap and not f32:
                # (a timer class whose launches are two symbols: the mean over its launches, one launch of each per step)
                names = [t.strip().replace(" ", "") for t in sym[dom].split(" + ")]
                traffic = sum(tj["kernels"][t]["hbm_bytes"] for t in names) / len(names)
                traffic_src = tfile + " (static: rocprofv3 --pmc passes of this command, not measured in this run)"
        except Exception:
            traffic = None
        roof = {"bound": "mfma", "achieved": achieved, "peak": PEAK, "unit": "TFLOP/s",
                "frac": achieved / PEAK, "traffic": traffic, "traffic_source": traffic_src,
                "kernel": sym[dom], "avg_launch_ms": dom_ms, "algorithmic_flops_per_launch": alg[dom],
                "launches_per_step": launches_per_step[dom], "overlap_level": args.overlap}
        # every strip product by the same definition, and the whole step: sum of algorithmic flops / ms_per_step
        per_kernel = {k: {"kernel": sym[k], "avg_launch_ms": per_launch[k], "algorithmic_flops_per_launch": alg[k],
                          "launches_per_step": launches_per_step[k],
                          "achieved": (alg[k] / (per_launch[k] * 1e-3) / 1e12) if per_launch[k] > 0 else 0.0}
                      for k in alg}
        for v in per_kernel.values():
            v["frac"] = v["achieved"] / PEAK
        step_flops = sum(alg[k] * timers[k][1] for k in alg) / float(args.steps)
        step_tf = step_flops / (elapsed / args.steps) / 1e12
        kuf = {}
        for name, mm in (("kuf_build", 0), ("kuf_build_sm", args.partials)):
            ms, n = timers[name]
            if n:
                # one launch builds the Kuf strips of a whole kernel family (all P activation or all P component GPs)
                gps = max(1, n_act_local if name == "kuf_build" else n_com_local)
                byts = gps * (T * (float(M) * N + N + M) + T * 2.0 * mm * (M + N))
                a = byts / (ms / n * 1e-3) / 1e9
                kuf[name] = {"bound": "hbm", "achieved": a, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": a / PEAK_HBM_GBS,
                             "avg_launch_ms": ms / n, "algorithmic_bytes_per_launch": byts, "latent_gps_per_launch": gps}
        # Kuf build of the spectral-mixture family, both rooflines (HBM-bound only for m <~ 13): the
        # float64 arithmetic of one entry is 4m (feature dot, on the matrix cores) + ~25 (distance, sqrt, exp) flops
        if "kuf_build_sm" in kuf:
            k = kuf["kuf_build_sm"]
            fl = k["latent_gps_per_launch"] * float(M) * N * (4.0 * args.partials + 25.0)
            k["partials"] = args.partials
            k["compute_roofline"] = {"bound": "f64 (vector + matrix share the DP units)", "algorithmic_flops_per_launch": fl,
                                     "achieved": fl / (k["avg_launch_ms"] * 1e-3) / 1e12, "peak": PEAK_F64_MFMA_TFLOPS,
                                     "unit": "TFLOP/s", "frac": fl / (k["avg_launch_ms"] * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS}
        if kuf_alone is not None:
            for name in ("kuf_build", "kuf_build_sm"):
                ms, n = kuf_alone[name]
                if n and name in kuf:
                    byts = kuf[name]["algorithmic_bytes_per_launch"]
                    al = {"avg_launch_ms": ms / n, "achieved": byts / (ms / n * 1e-3) / 1e9,
                          "frac": byts / (ms / n * 1e-3) / 1e9 / PEAK_HBM_GBS, "overlap_level": 0}
                    if "compute_roofline" in kuf[name]:
                        fl = kuf[name]["compute_roofline"]["algorithmic_flops_per_launch"]
                        al["compute_frac"] = fl / (ms / n * 1e-3) / 1e12 / PEAK_F64_MFMA_TFLOPS
                    kuf[name]["alone"] = al
        if kuf_m5 is not None:
            kuf["kuf_build_sm_m5"] = kuf_m5
        out = {
            "metric": "ELBO-steps/sec", "value": (1 if pitch else world) * args.steps / elapsed, "unit": "steps/s",
            "n_gpus": world, "backend": (args.backend if dist is not None else None),
            "rccl_ranks": (dist.get_world_size() if (dist is not None and args.backend == "nccl") else 0),
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if pitch else "weak", "vs_baseline": None, "dtype": args.float_type, "data": "synthetic",
            # (kept short: the driver's field cuts at ~100 characters; the long form is "workload_detail")
            "config": {"workload": "pdgp ELBO step N=%d M=%d P=%d m=%d %s full batch (no index draw), %s"
                                   % (N, M, args.P, args.partials, "f32 strips" if f32 else "f64",
                                      ("pitch-sharded" if args.shard == "pitch" else "gp-sharded") if pitch else "window/GPU"),
                       "workload_detail": "pdgp ELBO step (fwd + grad + Adam), N=%d frames x M=%d inducing x P=%d pitches "
                                          "(2P=%d latent GPs), m=%d partials, %s, full batch; %s"
                                          % (N, M, args.P, 2 * args.P, args.partials,
                                             "float32 strips and strip products (float64 Kuu / reductions)" if f32 else "float64",
                                             ("one model pitch-sharded over the GPUs (all-reduce of 3N+1 doubles per step)"
                                              if args.shard == "pitch" else
                                              "one model, its 2P latent GPs dealt over the GPUs (all-gather of (fmean, fvar) per step)")
                                             if pitch else "one independent window per GPU"),
                       "N": N, "M": M, "P": args.P, "partials": args.partials, "whiten": True,
                       "parallelism": (("pitch-sharded x%d" if args.shard == "pitch" else "gp-sharded x%d") if pitch
                                       else "window-per-gpu x%d") % world,
                       "overlap_level": args.overlap},
            "roofline": roof,
            "mfma_frac_step": {"algorithmic_flops_per_step": step_flops, "achieved": step_tf, "unit": "TFLOP/s",
                               "peak": PEAK, "frac": step_tf / PEAK},
            "roofline_strips": per_kernel,
            "roofline_kuf_build": kuf,
            "kernel_ms_per_step": {k: ms / args.steps for k, (ms, n) in timers.items()},
            "elbo_final": elbo_final,
        }
        if forward_only is not None:
            out["forward_only"] = forward_only
        if extra_pitch is not None:
            out["pitch_sharded"] = extra_pitch
        if extra_gp is not None:
            out["gp_sharded"] = extra_gp
        if cfg3 is not None:
            out["cfg3_fp32"] = cfg3
        if world == 1 and not args.no_sgpr_lines and not args.no_f32_line and (args.N, args.M, args.P) == (32768, 512, 12):
            # the other two workloads of the path, same run: BASELINE configs[4] and the reference's window loop
            model = res = None
            torch.cuda.empty_cache()
            out["cfg5_sgpr"] = bench_cfg5(args, not args.no_cpu)
            torch.cuda.empty_cache()
            out["windows"] = bench_windows(args, not args.no_cpu)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args)
            out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
