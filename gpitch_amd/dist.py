"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
MI355X node; "gloo" for the CPU rehearsal in tests).

The reference has no distributed code.  It scales N by independent windows / segments
(gpitch/window_overlap.py:7-16,194-211) that are fitted one after another
(gpitch/transcription.py:265-288, gpitch/separation.py:279-313): windows are the sharding unit, there
is no data-path exchange, and only the scalar total ELBO is reduced for reporting.
"""
import os


def env_world():
    return int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))


def window_assignment(num_windows, world_size, rank):
    """Round-robin windows over ranks (window w -> rank w mod world)."""
    return list(range(rank, num_windows, world_size))


def init_process_group(backend=None):
    import torch
    import torch.distributed as dist
    world, rank, local_rank = env_world()
    if world == 1 or dist.is_initialized():
        return dist if world > 1 else None
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend, **kw)
    return dist


def allreduce_sum_(t):
    """in-place sum over ranks of a (device or host) tensor; no-op without a process group.  (A one-rank group still goes
    through the backend: that is how a one-GPU box exercises the RCCL path, stream ordering included.)"""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        backend = dist.get_backend()
        if t.is_cuda and backend == "gloo":
            # CPU rehearsal backend: stage through the host (RCCL reduces device memory directly)
            host = t.cpu()
            dist.all_reduce(host)
            t.copy_(host)
        elif (not t.is_cuda) and backend == "nccl":
            # a host tensor on the production group (nccl only: "No backend type associated with device type cpu"):
            # reduce a device copy
            import torch
            dev = t.to(torch.device("cuda", torch.cuda.current_device()))
            dist.all_reduce(dev)
            t.copy_(dev.cpu())
        else:
            dist.all_reduce(t)
    return t


def require_group(world, what):
    """a sharded model's cross-rank step needs an initialised process group of exactly `world` ranks: without one the
    sum over ranks would silently be the local rows alone"""
    import torch.distributed as dist
    if world <= 1:
        return
    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("%s: the model is sharded over %d ranks but torch.distributed is not initialised" % (what, world))
    if dist.get_world_size() != world:
        raise RuntimeError("%s: the model is sharded over %d ranks, the process group has %d" % (what, world, dist.get_world_size()))


def allreduce_max_(t):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


# ---------------------------------------------------------------------------------------------------------------
# GP-sharded Pdgp (SURVEY section 8e option 2): the 2 P latent GPs of ONE model dealt over the ranks, one all-gather of
# (fmean, fvar) per step.  Everything below is plain torch / Python, device-agnostic: the gloo tests run it on the CPU.
def gp_assignment(num_gps, world_size, rank):
    """latent GPs (rows of the model's order [g_0..g_{P-1}, f_0..f_{P-1}], pdgp.py:157-164) held by `rank`: g = rank (mod world)"""
    return list(range(rank, num_gps, world_size))


def gp_exchange_layout(num_gps, world_size, n):
    """(rows per rank, doubles per rank) of the all-gather's send block: [fmean rows | fvar rows | KL sum + 7 pad].  Every
    rank sends the same size (ranks with one GP fewer leave their last row slot zero)."""
    per = -(-num_gps // world_size)
    return per, 2 * per * n + 8


def gp_assemble(gathered, num_gps, world_size, n):
    """gathered: (world * block,) — the all-gather's output.  Returns (fmean (num_gps, n), fvar (num_gps, n), kl_total (1,))
    in the model's row order: rank r's l-th row is latent GP r + l * world."""
    per, blk = gp_exchange_layout(num_gps, world_size, n)
    r = gathered.view(world_size, blk)
    fm = r[:, :per * n].reshape(world_size, per, n).permute(1, 0, 2).reshape(per * world_size, n)[:num_gps].contiguous()
    fv = r[:, per * n:2 * per * n].reshape(world_size, per, n).permute(1, 0, 2).reshape(per * world_size, n)[:num_gps].contiguous()
    kl = r[:, 2 * per * n].sum().reshape(1)
    return fm, fv, kl


def allgather_(out, send):
    """out (world * len(send),) <- every rank's `send`, rank-major; a plain copy without a process group (a one-rank group
    still goes through the backend, as allreduce_sum_ does)."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        if send.is_cuda and dist.get_backend() == "gloo":
            host_out, host_in = out.cpu(), send.cpu()
            dist.all_gather_into_tensor(host_out, host_in)
            out.copy_(host_out)
        else:
            dist.all_gather_into_tensor(out, send)
    else:
        out.copy_(send)
    return out
