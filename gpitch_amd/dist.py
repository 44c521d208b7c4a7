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
        if t.is_cuda and dist.get_backend() == "gloo":
            # CPU rehearsal backend: stage through the host (RCCL reduces device memory directly)
            host = t.cpu()
            dist.all_reduce(host)
            t.copy_(host)
        else:
            dist.all_reduce(t)
    return t


def allreduce_max_(t):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t
