"""World-size-2 rehearsal (gloo, CPU) of the multi-GPU path: windows are sharded across ranks with no
data-path collective and the scalar ELBO is all-reduced (bench.py / gpitch_amd/dist.py).  The per-window
ELBO is the oracle's here (no GPU in this container); the sharding and reduction logic is the product's."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _window_elbo(w):
    from gpitch_amd.synth import make_problem
    from helpers import oracle_elbo
    return float(oracle_elbo(make_problem(128, 8, 1, num_partials=2, seed=w)))


def _worker(rank, world, port, nwin, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gpitch_amd import dist as gd
    d = gd.init_process_group("gloo")
    mine = gd.window_assignment(nwin, world, rank)
    t = torch.tensor([sum(_window_elbo(w) for w in mine), float(len(mine))], dtype=torch.float64)
    gd.allreduce_sum_(t)
    tm = torch.tensor([float(rank + 1)], dtype=torch.float64)
    gd.allreduce_max_(tm)
    if rank == 0:
        out.put((t[0].item(), t[1].item(), tm.item()))
    d.destroy_process_group()


def test_window_sharding_and_scalar_allreduce_gloo():
    nwin, world = 5, 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, nwin, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    total, count, tmax = q.get()
    serial = sum(_window_elbo(w) for w in range(nwin))
    assert count == nwin and tmax == world
    assert abs(total - serial) <= 1e-12 * abs(serial)


def test_window_assignment_covers_all():
    from gpitch_amd.dist import window_assignment
    for nwin, world in [(8, 8), (261, 8), (3, 4), (12, 5)]:
        got = sorted(sum((window_assignment(nwin, world, r) for r in range(world)), []))
        assert got == list(range(nwin))


# ---------------------------------------------------------------------------------------------
# Pitch-sharded single model: ranks hold disjoint pitch subsets and exchange ONE all-reduce of
# the per-frame sums [A | B | D] plus the KL slot.  The per-rank arithmetic here is the oracle's (no GPU in this
# container); the assignment, the exchange layout and the reduction are the product's.
def _pitch_partial(prob, pitches):
    from oracle import gpflow05 as orc
    from oracle.backend import NP
    n = prob["x"].shape[0]
    A = np.zeros(n); B = np.zeros(n); D = np.zeros(n); kl = 0.0
    for p in pitches:
        mg, vg = orc.conditional(prob["x"], prob["za"][p], prob["kern_act"][p], prob["q_mu_act"][p],
                                 prob["q_sqrt_act"][p], whiten=True, xp=NP)
        mf, vf = orc.conditional(prob["x"], prob["zc"][p], prob["kern_com"][p], prob["q_mu_com"][p],
                                 prob["q_sqrt_com"][p], whiten=True, xp=NP)
        E1, E2 = orc.hermgauss1d(mg, vg, 20, orc.nlinfun(0), xp=NP)
        a = (E1 * mf).reshape(-1)
        A += a; D += a * a
        B += (E2 * (vf + mf * mf)).reshape(-1)
        kl += float(orc.gauss_kl(prob["q_mu_act"][p], prob["q_sqrt_act"][p], xp=NP))
        kl += float(orc.gauss_kl(prob["q_mu_com"][p], prob["q_sqrt_com"][p], xp=NP))
    return np.concatenate([A, B, D, [kl]])


def _elbo_from_exchange(prob, xchg):
    n = prob["x"].shape[0]
    A, B, D, kl = xchg[:n], xchg[n:2 * n], xchg[2 * n:3 * n], xchg[3 * n]
    y = prob["y"].reshape(-1)
    s2 = float(prob["noise_var"])
    ve = -0.5 * ((y * y - 2 * y * A + B + (A * A - D)) / s2 + np.log(2 * np.pi) + np.log(s2))
    return ve.sum() - kl


def _pitch_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gpitch_amd import dist as gd
    from gpitch_amd.pdgp import pitch_assignment
    from gpitch_amd.synth import make_problem
    d = gd.init_process_group("gloo")
    prob = make_problem(160, 8, 3, num_partials=2, seed=5)
    t = torch.as_tensor(_pitch_partial(prob, pitch_assignment(3, world, rank)))
    gd.allreduce_sum_(t)
    out.put((rank, float(_elbo_from_exchange(prob, t.numpy()))))
    d.destroy_process_group()


def test_pitch_sharded_exchange_gloo():
    from gpitch_amd.synth import make_problem
    from helpers import oracle_elbo
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_pitch_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get() for _ in range(world))
    ref = float(oracle_elbo(make_problem(160, 8, 3, num_partials=2, seed=5)))
    assert got[0] == got[1]                     # every rank derives the same ELBO from the reduced vector
    assert abs(got[0] - ref) <= 1e-11 * abs(ref)


def test_pitch_assignment_covers_all():
    from gpitch_amd.pdgp import pitch_assignment
    for P, world in [(12, 8), (12, 2), (5, 5), (88, 8)]:
        got = sorted(sum((pitch_assignment(P, world, r) for r in range(world)), []))
        assert got == list(range(P))
    assert [len(pitch_assignment(12, 8, r)) for r in range(8)] == [2, 2, 2, 2, 1, 1, 1, 1]


# ---------------------------------------------------------------------------------------------
# Frame-sharded SGPRSS window: ranks hold frame slices and exchange [H | u | sum y^2 | tr H].
# Per-rank arithmetic is numpy here (no GPU in this container); the exchange layout, the frame split and the
# reduction are the product's (gpitch_amd/sgpr_ss.py:_frames, include/gpitch_abi.h gp_sgpr_bound_begin/_end).
def _sgpr_problem():
    rng = np.random.RandomState(3)
    N, M = 257, 12
    X = np.linspace(0, 0.02, N).reshape(-1, 1)
    Y = np.sin(2 * np.pi * 220. * X) + 0.1 * rng.randn(N, 1)
    Z = X[::N // M][:M].copy()
    kl = [{"type": "mercer_matern12sm", "variance": 1.0, "lengthscales": 0.05, "energy": [0.6, 0.4],
           "frequency": [220., 440.]},
          {"type": "matern32", "variance": 0.5, "lengthscales": 0.01, "energy": [], "frequency": []}]
    return X, Y, Z, kl, 0.3


def _sgpr_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gpitch_amd import dist as gd
    from gpitch_amd.sgpr_ss import SGPRSS
    from oracle import gpflow05 as orc
    d = gd.init_process_group("gloo")
    X, Y, Z, kl, s2 = _sgpr_problem()
    M = Z.shape[0]
    probe = SGPRSS.__new__(SGPRSS)                       # only the frame-split rule of the product is needed here
    object.__setattr__(probe, "_shard", (rank, world))
    probe.__dict__["X"] = type("D", (), {"shape": X.shape})()
    fr = probe._frames()
    Xs, Ys = X[fr], Y[fr]
    L = np.linalg.cholesky(orc.K_sum(kl, Z) + 1e-6 * np.eye(M))
    A = np.linalg.solve(L, orc.K_sum(kl, Z, Xs))          # A' = L^-1 Kuf (not yet divided by sigma)
    xchg = np.concatenate([(A @ A.T).reshape(-1), (A @ Ys).reshape(-1), [float((Ys ** 2).sum())], [float((A * A).sum())]])
    t = torch.as_tensor(xchg)
    gd.allreduce_sum_(t)
    v = t.numpy()
    H, u, yy, trH = v[:M * M].reshape(M, M), v[M * M:M * M + M], v[M * M + M], v[M * M + M + 1]
    N = X.shape[0]
    LB = np.linalg.cholesky(H / s2 + np.eye(M))
    c = np.linalg.solve(LB, u) / s2
    kd = float(orc.Kdiag_sum(kl, X[:1])[0])
    bound = (-0.5 * N * np.log(2 * np.pi) - np.log(np.diag(LB)).sum() - 0.5 * N * np.log(s2) - 0.5 * yy / s2
             + 0.5 * (c ** 2).sum() - 0.5 * N * kd / s2 + 0.5 * trH / s2)
    out.put((rank, float(bound), fr.start, fr.stop))
    d.destroy_process_group()


def test_frame_sharded_sgpr_exchange_gloo():
    from oracle import gpflow05 as orc
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_sgpr_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = sorted(q.get() for _ in range(world))
    X, Y, Z, kl, s2 = _sgpr_problem()
    ref = float(orc.sgpr_bound(X, Y, Z, kl, s2))
    assert got[0][1] == got[1][1]
    assert abs(got[0][1] - ref) <= 1e-10 * abs(ref)
    assert (got[0][2], got[0][3], got[1][2], got[1][3]) == (0, 129, 129, 257)     # contiguous, sizes differ by <= 1


# ---------------------------------------------------------------------------------------------
# GP-sharded single model (SURVEY §8e option 2): ranks hold disjoint sets of the 2P LATENT GPs and exchange ONE all-gather
# of (fmean, fvar) per GP plus the KL scalar; every rank then evaluates the whole likelihood.  The per-GP arithmetic here is
# the oracle's (no GPU in this container); the assignment, the send-block layout, the all-gather and the re-assembly into
# the model's row order are the product's (gpitch_amd/dist.py — the same functions Pdgp(shard=("gp", r, w)) calls).
def _gp_send_block(prob, rank, world):
    from oracle import gpflow05 as orc
    from oracle.backend import NP
    from gpitch_amd import dist as gd
    P, n = prob["P"], prob["x"].shape[0]
    per, blk = gd.gp_exchange_layout(2 * P, world, n)
    send = np.zeros(blk)
    for l, g in enumerate(gd.gp_assignment(2 * P, world, rank)):
        act, i = (True, g) if g < P else (False, g - P)
        z, k, qm, qs = ((prob["za"], prob["kern_act"], prob["q_mu_act"], prob["q_sqrt_act"]) if act else
                        (prob["zc"], prob["kern_com"], prob["q_mu_com"], prob["q_sqrt_com"]))
        m, v = orc.conditional(prob["x"], z[i], k[i], qm[i], qs[i], whiten=True, xp=NP)
        send[l * n:(l + 1) * n] = m.reshape(-1)
        send[(per + l) * n:(per + l + 1) * n] = v.reshape(-1)
        send[2 * per * n] += float(orc.gauss_kl(qm[i], qs[i], xp=NP))
    return send


def _gp_worker(rank, world, port, P, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from gpitch_amd import dist as gd
    from gpitch_amd.synth import make_problem
    from oracle import gpflow05 as orc
    d = gd.init_process_group("gloo")
    prob = make_problem(160, 8, P, num_partials=2, seed=6)
    n = prob["x"].shape[0]
    send = torch.as_tensor(_gp_send_block(prob, rank, world))
    recv = torch.zeros(send.numel() * world, dtype=torch.float64)
    gd.allgather_(recv, send)
    fm, fv, kl = gd.gp_assemble(recv, 2 * P, world, n)
    ve = orc.mpd_variational_expectations(fm.numpy().T.copy(), fv.numpy().T.copy(), prob["y"], prob["noise_var"], P)
    out.put((rank, float(ve.sum() - kl.item())))
    d.destroy_process_group()


def _run_gp_world(world, P, port_base):
    from gpitch_amd.synth import make_problem
    from helpers import oracle_elbo
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    port = port_base + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gp_worker, args=(r, world, port, P, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    got = dict(q.get() for _ in range(world))
    ref = float(oracle_elbo(make_problem(160, 8, P, num_partials=2, seed=6)))
    assert sorted(got) == list(range(world))
    for r in range(world):
        assert abs(got[r] - ref) <= 1e-11 * abs(ref), (r, got[r], ref)        # every rank holds the whole model's ELBO


def test_gp_sharded_allgather_exchange_gloo_world2():
    _run_gp_world(2, 3, 33500)


def test_gp_sharded_allgather_exchange_gloo_world3_ragged():
    """2P = 4 latent GPs on 3 ranks: rank 0 holds two (g_0 and f_1), ranks 1 and 2 one each and a padded slot"""
    _run_gp_world(3, 2, 35500)


def test_gp_assignment_and_layout():
    from gpitch_amd.dist import gp_assignment, gp_exchange_layout, gp_assemble
    for G, world in [(24, 8), (24, 5), (4, 3), (10, 10), (6, 1)]:
        got = sorted(sum((gp_assignment(G, world, r) for r in range(world)), []))
        assert got == list(range(G))
        assert max(len(gp_assignment(G, world, r)) for r in range(world)) == gp_exchange_layout(G, world, 7)[0]
    # 24 GPs on 8 ranks: three each (the pitch-sharded form has 2,2,2,2,1,1,1,1 pitches = 4,4,4,4,2,2,2,2 GPs)
    assert [len(gp_assignment(24, 8, r)) for r in range(8)] == [3] * 8
    # re-assembly: rank r's l-th row is GP r + l * world
    G, world, n = 5, 2, 3
    per, blk = gp_exchange_layout(G, world, n)
    buf = torch.zeros(world * blk, dtype=torch.float64)
    for r in range(world):
        for l, g in enumerate(gp_assignment(G, world, r)):
            buf[r * blk + l * n:r * blk + (l + 1) * n] = 100 + g
            buf[r * blk + (per + l) * n:r * blk + (per + l + 1) * n] = 200 + g
        buf[r * blk + 2 * per * n] = 0.5 + r
    fm, fv, kl = gp_assemble(buf, G, world, n)
    assert fm.shape == (G, n) and torch.equal(fm[:, 0], 100 + torch.arange(G, dtype=torch.float64))
    assert torch.equal(fv[:, 2], 200 + torch.arange(G, dtype=torch.float64)) and kl.item() == 2.0
