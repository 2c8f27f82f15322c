"""The multi-GPU paths over a real RCCL communicator with more than one rank: skipped unless two
GPUs are visible.  One process per GPU (torch.multiprocessing.spawn), backend "nccl":

  * svicp_shard_run: a pair's source split over two ranks, target replicated, one all-reduce of
    the fixed-point sums per iteration inside the library, against the one-GPU svicp_run;
  * run_pairs_split: independent pairs split over two ranks and gathered, against one-by-one runs.

The reference is single-process (PointCloud/Registration/test.cpp:14-29): what is checked is that
splitting changes nothing but the fp64 rounding inside the tile sums (the ranks cut the source
into other tiles), i.e. X within 1e-10, and bit-equal results for the pairs."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch
    return torch.cuda.device_count()      # (counting devices does not initialise the GPU)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import stereovision_amd as sv
    from stereovision_amd import synth
    from stereovision_amd.batched import HipBatchEngine, pair_bounds, run_pairs_split
    from stereovision_amd.sharded import init_comm, run_sharded_native, shard_bounds
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        # ---- one pair, source split over the ranks, the loop inside the library ----
        tgt, nrm, src, _ = synth.make_pair(256, pair=1)
        b, e = shard_bounds(len(src), world, rank)
        with sv.Context(rank) as c:
            c.set_target(tgt, nrm)
            c.set_source(np.ascontiguousarray(src[b:e]))
            init_comm(c)
            T, info = run_sharded_native(c, 12)
            captured, comm_world = c.shard_run_mode()
            X, I = c.history()
            # a second run after only THIS rank set its part again: the ranks must still make the
            # same collective calls (the agreement on the total does not depend on one rank's history)
            if rank == 0:
                c.set_source(np.ascontiguousarray(src[b:e]))
            T2, _ = run_sharded_native(c, 12)
            moved = c.get_source()
            c.comm_destroy()
        np.savez(os.path.join(out_dir, f"shard{rank}.npz"), T=T, T2=T2, X=X, inliers=info.inliers,
                 captured=int(captured), world=comm_world, moved=moved)
        # ---- independent pairs split over the ranks, gathered in pair order ----
        count = 5
        first, last = pair_bounds(count, world, rank)
        tg, nr, sr = [], [], []
        for pid in range(first, last):
            t, n, s, _ = synth.make_pair(64, pair=pid, pose=synth.small_pose(pid))
            tg.append(t); nr.append(n); sr.append(s)
        eng = HipBatchEngine(tg, nr, sr, device=rank)
        Tall, infos = run_pairs_split(eng, count, 10)
        eng.close()
        np.savez(os.path.join(out_dir, f"pairs{rank}.npz"), T=Tall, info=infos)
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(_gpus() < 2, reason="needs two GPUs: one rank per GPU over RCCL")
def test_two_ranks_over_rccl_equal_one_gpu(tmp_path):
    import torch.multiprocessing as mp
    import stereovision_amd as sv
    from stereovision_amd import synth
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)

    tgt, nrm, src, _ = synth.make_pair(256, pair=1)
    with sv.Context(0) as full:
        full.set_target(tgt, nrm)
        full.set_source(src)
        T_full, info_full = full.run(12)
        X_full, _ = full.history()
        moved_full = full.get_source()
    parts = [np.load(tmp_path / f"shard{r}.npz") for r in range(2)]
    for p in parts:
        assert int(p["world"]) == 2
        assert np.abs(p["X"] - X_full).max() < 1e-10
        assert np.abs(p["T"] - T_full).max() < 1e-10
        assert p["inliers"] == info_full.inliers
    # every rank solved from the same integers: the same bits on both, run after run
    assert np.array_equal(parts[0]["T"], parts[1]["T"]) and np.array_equal(parts[0]["X"], parts[1]["X"])
    assert np.array_equal(parts[0]["T"], parts[0]["T2"]) and np.array_equal(parts[1]["T"], parts[1]["T2"])
    moved = np.concatenate([p["moved"] for p in parts])
    assert np.abs(moved - moved_full).max() <= 4e-5
    print("collective captured in the graph:", [int(p["captured"]) for p in parts])

    pairs = [np.load(tmp_path / f"pairs{r}.npz") for r in range(2)]
    assert np.array_equal(pairs[0]["T"], pairs[1]["T"]) and np.array_equal(pairs[0]["info"], pairs[1]["info"])
    for pid in range(5):
        t, n, s, _ = synth.make_pair(64, pair=pid, pose=synth.small_pose(pid))
        with sv.Context(0) as c:
            c.set_target(t, n)
            c.set_source(s)
            T_one, info_one = c.run(10)
        assert np.array_equal(pairs[0]["T"][pid], T_one), pid
        assert pairs[0]["info"][pid, 1] == info_one.inliers
