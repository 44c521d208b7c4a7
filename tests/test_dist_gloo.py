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
