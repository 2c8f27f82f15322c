"""Batches of independent cloud pairs split over ranks (one process per GPU).

Pair i of B goes to rank g with g * B / G <= i < (g + 1) * B / G; a rank registers its pairs
in one arena batch (one kernel launch per iteration for all of them); nothing is exchanged
while the iterations run.  Afterwards the B x 16 doubles of the transforms are gathered on
every rank.  The reference registers one pair per CPoint2PlaneICP::run call
(PointCloud/Registration/test.cpp:22-25) in one process; a caller with many pairs loops.
"""
import numpy as np
import torch.distributed as dist


def pair_bounds(count, world_size, rank):
    """[first, last) of the pairs rank `rank` registers."""
    count, world_size, rank = int(count), int(world_size), int(rank)
    return rank * count // world_size, (rank + 1) * count // world_size


class HipBatchEngine:
    """A rank's pairs on its GPU: an ArenaBatch over targets[first:last] etc."""

    def __init__(self, targets, normals, sources, device=0, **options):
        from .icp import ArenaBatch
        self.count = len(targets)
        self.batch = ArenaBatch(targets, normals, sources, device=device, **options) if self.count else None

    def run(self, iterations):
        """Returns (count x 4 x 4 transforms, count x 2 {status, inliers})."""
        if self.batch is None:
            return np.zeros((0, 4, 4)), np.zeros((0, 2))
        T, infos = self.batch.run(iterations)
        return T, np.array([[i.status, i.inliers] for i in infos], np.float64)

    def close(self):
        if self.batch is not None:
            self.batch.close()
            self.batch = None


def run_pairs_split(engine, count, iterations, group=None):
    """Registers this rank's pairs with `engine` (built over pair_bounds(count, world, rank))
    and gathers every rank's results.  Returns (count x 4 x 4, count x 2) in pair order, the
    same on every rank."""
    T, info = engine.run(int(iterations))
    T = np.ascontiguousarray(T, np.float64).reshape(-1, 4, 4)
    info = np.ascontiguousarray(info, np.float64).reshape(-1, 2)
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        assert len(T) == count
        return T, info
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    first, last = pair_bounds(count, world, rank)
    assert len(T) == last - first, "the engine must hold exactly this rank's pairs"
    parts = [None] * world
    dist.all_gather_object(parts, (T, info), group=group)      # B x 18 doubles in all, once per run
    for r, (t, i) in enumerate(parts):
        f, l = pair_bounds(count, world, r)
        assert len(t) == l - f
    return (np.concatenate([t for t, _ in parts]).reshape(count, 4, 4),
            np.concatenate([i for _, i in parts]).reshape(count, 2))
