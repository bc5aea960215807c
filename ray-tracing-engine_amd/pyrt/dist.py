"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by 8x8-pixel tiles (rt_params.rank/world/tile): every rank
integrates ALL samples of the pixels it owns, in order, into a zero-initialised
full-frame accumulator {sum r, g, b, primary-hit count}.  The only exchange step
is one SUM reduce of that accumulator to rank 0 per frame: each pixel is non-zero
on exactly one rank and adding zeros is exact, so the N-GPU frame is bit-identical
to the 1-GPU frame.  Message size is w*h*16 B (16 MiB at 1024^2, 64 MiB at 2048^2):
one ring step over one ~153 GB/s xGMI link is well under a millisecond, which is
why nothing more elaborate than a single reduce is used.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend):
    """Returns (rank, world, local_rank); initialises the process group if WORLD_SIZE > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def active():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def barrier():
    if active():
        dist.barrier()


def reduce_frame(accum, dst=0):
    """In-place SUM reduce of the per-rank accumulator onto rank `dst`."""
    if active():
        if accum.is_cuda and dist.get_backend() == "gloo":
            # rehearsal mode (several ranks sharing one GPU): gloo reduces on the host
            host = accum.cpu()
            dist.reduce(host, dst=dst, op=dist.ReduceOp.SUM)
            accum.copy_(host)
        else:
            dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM)
    return accum


def max_over_ranks(value, device):
    if active() and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device):
    if active() and dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([int(v) for v in values], dtype=torch.int64, device=device)
    if active():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


def shutdown():
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
