"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" =
RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The path shards by 8x8-pixel tiles (rt_params.rank/world/tile): every rank
integrates ALL samples of the pixels it owns, in order, into a zero-initialised
full-frame accumulator {sum r, g, b, primary-hit count}.  The only exchange step
is the assembly of the frame on rank 0: FrameAssembler sends each rank's OWNED
granules to rank 0 — point to point, exactly counts[r] granules (w*h*16/N bytes per
rank: 2 MiB at 1024^2 and 8 ranks, each over its own xGMI link to rank 0; one batch of
isend / irecv = one ncclGroupStart ... ncclGroupEnd under RCCL) — so the N-GPU frame is
bit-identical to the 1-GPU frame by construction: pixels are copied, never summed.  reduce_frame (a full-frame SUM reduce of mostly
zeros, exact for the same reason) is kept as the reference exchange for the tests.
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


def owned_granule_index(width, height, rank, world, tile):
    """Host statement of the owned-granule order (csrc/rt_api.cpp owned_granules): flat pixel
    indices [n_granules][64] of rank's 8x8 granules, row-major; -1 outside the image."""
    import numpy as np
    gx, gy = (width + 7) // 8, (height + 7) // 8
    y8, x8 = np.meshgrid(np.arange(gy), np.arange(gx), indexing="ij")
    own = np.ones_like(x8, bool) if world <= 1 else ((x8 * 8 // tile + y8 * 8 // tile) % world == rank)
    gxs, gys = x8[own], y8[own]  # row-major order of the boolean mask
    ly, lx = np.divmod(np.arange(64), 8)
    px = gxs[:, None] * 8 + lx[None, :]
    py = gys[:, None] * 8 + ly[None, :]
    idx = py * width + px
    idx[(px >= width) | (py >= height)] = -1
    return idx


class FrameAssembler:
    """Assembles the tile-sharded frame on rank 0 by moving only OWNED pixels: every rank
    packs its 8x8 granules ([granule][64] float4), one batch of point-to-point sends brings
    them to rank 0 (exactly counts[r] granules from rank r: 1/N of the frame per rank instead
    of a full-frame SUM reduce of mostly zeros, and nothing padded to the largest share),
    rank 0 scatters them.  On a GPU the pack/scatter are the library's kernels (rt_pack_owned_device /
    rt_unpack_owned_device); on CPU tensors (gloo tests) the same order in numpy."""

    def __init__(self, ctx, params, rank, world, device):
        self.ctx, self.params, self.rank, self.world = ctx, params, rank, world
        self.w, self.h, self.tile = params.width, params.height, params.tile or 8
        self.cuda = torch.device(device).type == "cuda"
        self.backend = dist.get_backend() if active() else None
        if world <= 1:
            return
        if ctx is not None and self.cuda:
            import pyrt
            self.counts = [pyrt.owned_granules(params, r) for r in range(world)]
        else:
            self.index = [owned_granule_index(self.w, self.h, r, world, self.tile) for r in range(world)]
            self.counts = [len(ix) for ix in self.index]
        # gloo cannot move device tensors: stage through the host in rehearsal mode
        self.stage_dev = "cpu" if (self.cuda and self.backend == "gloo") else device
        self.packed = torch.zeros((self.counts[rank] * 64, 4), dtype=torch.float32, device=device)
        self.recv = ([None] + [torch.zeros((self.counts[r] * 64, 4), dtype=torch.float32, device=self.stage_dev) for r in range(1, world)]
                     if rank == 0 else None)

    def assemble(self, accum, stream=0):
        """Every rank must call this the same number of times.  A failure (pack kernel, gather) is
        raised, never papered over per rank: a rank that switched to another collective on its own
        would leave the others waiting in the gather."""
        if self.world <= 1:
            return accum
        return self._gather(accum, stream)

    def _gather(self, accum, stream):
        if self.cuda:
            self.ctx.pack_owned(self.params, accum.data_ptr(), self.packed.data_ptr(), stream)
            send = self.packed if self.stage_dev != "cpu" else self.packed.cpu()
        else:
            ix = torch.from_numpy(self.index[self.rank].reshape(-1).clip(min=0))
            self.packed[: len(ix)] = accum.view(-1, 4)[ix]
            send = self.packed
        # rank r > 0 -> rank 0, counts[r] granules each (rank 0's own granules are already in place)
        if self.rank == 0:
            ops = [dist.P2POp(dist.irecv, self.recv[r], r) for r in range(1, self.world) if self.counts[r]]
        else:
            ops = [dist.P2POp(dist.isend, send, 0)] if self.counts[self.rank] else []
        if ops:
            for req in dist.batch_isend_irecv(ops):
                req.wait()
        self.calls = getattr(self, "calls", 0) + 1
        if self.rank == 0:
            for r in range(1, self.world):
                if not self.counts[r]:
                    continue
                if self.cuda:
                    buf = self.recv[r] if self.stage_dev != "cpu" else self.recv[r].to(accum.device)
                    self.ctx.unpack_owned(self.params, r, buf.data_ptr(), accum.data_ptr(), stream)
                    if self.stage_dev == "cpu":
                        torch.cuda.synchronize()  # buf is a temporary
                else:
                    ix = self.index[r].reshape(-1)
                    ok = torch.from_numpy(ix >= 0)
                    accum.view(-1, 4)[torch.from_numpy(ix[ix >= 0])] = self.recv[r][: len(ix)][ok]
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
