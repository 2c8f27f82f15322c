"""Registration of one large cloud pair sharded over ranks (one process per GPU).

Every rank holds the whole target grid and a contiguous slice of the source.
Per iteration each rank produces 32 doubles: the change its slice made to the
sums of the normal equations (21 upper-triangle JtJ entries, 6 Jtr entries, 3
statistics, 2 pad; the library keeps the running totals).  The ranks all-reduce
that block (RCCL over xGMI when the process group's backend is "nccl"), and
every rank solves the same 6x6 system, so no broadcast follows.  This is the only
exchange step of CPoint2PlaneICP::singleRun (CPoint2PlaneICP.cpp:63-120);
the reference itself is single-process.
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n, world_size, rank):
    """Contiguous [begin, end) of rank's slice of n source points."""
    base, rem = divmod(int(n), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class HipShardEngine:
    """Runs a rank's slice on its GPU through the C ABI's svicp_shard_* calls."""

    def __init__(self, ctx, device):
        self.ctx = ctx
        self.device = torch.device(device)
        # the library's sums are 64-bit fixed-point numbers: integers as far as the exchange goes
        self.buf = torch.zeros(32, dtype=torch.int64, device=self.device)
        # the library launches on torch's current stream, so collectives that
        # torch orders after that stream see finished sums
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def begin(self, total_points=None, group=None):
        """total_points: the size of the whole source (every rank must pass the same number);
        None: asked of the process group (sum of the ranks' parts), or this rank's own part
        when there is no group."""
        if total_points is None:
            n = self.ctx.sizes()[1]
            if dist.is_initialized() and dist.get_world_size(group) > 1:
                t = torch.tensor([n], dtype=torch.int64)
                if dist.get_backend(group) == "nccl":
                    t = t.to(self.device)
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
                n = int(t.item())
            total_points = n
        self.ctx.shard_set_total(total_points)
        self.ctx.shard_begin()

    def accumulate(self):
        self.ctx.shard_accumulate(self.buf.data_ptr())
        return self.buf

    def update(self, summed):
        assert summed.is_cuda and summed.dtype == torch.int64 and summed.numel() == 32
        self.ctx.shard_update(summed.data_ptr())

    def finish(self):
        return self.ctx.shard_finish()


def all_reduce_sums(buf, group=None):
    """Sums the 32-double block over ranks, in place.  With a backend that only
    moves host memory (gloo) a device block is staged through the host."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return buf
    backend = dist.get_backend(group)
    if buf.is_cuda and backend != "nccl":
        host = buf.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        buf.copy_(host)
    else:
        dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf


def run_sharded(engine, iterations, group=None):
    """The iteration loop of CPoint2PlaneICP::run (CPoint2PlaneICP.cpp:31-57)
    over a sharded source.  Returns what engine.finish() returns on this rank:
    the accumulated 4x4 transform (identical on every rank) and run info."""
    engine.begin(group=group) if isinstance(engine, HipShardEngine) else engine.begin()
    for _ in range(int(iterations)):
        buf = engine.accumulate()
        buf = all_reduce_sums(buf, group)
        engine.update(buf)
    return engine.finish()


def init_comm(ctx, group=None):
    """Gives `ctx` an RCCL communicator over the ranks of `group` (or a 1-rank one without
    torch.distributed): rank 0 asks the library for the 128-byte identifier, torch.distributed
    carries it to the other ranks (any backend: it is host bytes), every rank joins."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [ctx.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0,
                                   group=group)
        ctx.comm_init(box[0], rank, world)
    else:
        ctx.comm_init(ctx.comm_unique_id(), 0, 1)


def run_sharded_native(ctx, iterations):
    """The same loop as run_sharded, inside the library (svicp_shard_run): iteration kernel
    and RCCL all-reduce per iteration on one stream, captured in a HIP graph; no Python and
    no host round trip per iteration.  `ctx` holds the whole target and this rank's part of
    the source; with more than one rank it needs init_comm first."""
    return ctx.shard_run(int(iterations))
