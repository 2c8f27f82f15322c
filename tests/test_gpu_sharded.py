"""GPU tests of the sharded path: two contexts on one device each hold half of
the source; their 32-double blocks are summed as the all-reduce would."""
import os
import socket
import sys

import numpy as np
import pytest

import refmath

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_shards_in_one_process_match_unsharded(po):
    import torch
    import stereovision_amd as sv
    from stereovision_amd import synth
    from stereovision_amd.sharded import HipShardEngine, shard_bounds
    tgt, nrm, src, _ = synth.make_pair(128)
    full = sv.Context(0)
    full.set_target(tgt, nrm)
    full.set_source(src)
    T_full, info_full = full.run(5)

    engines = []
    for r in range(2):
        c = sv.Context(0)
        c.set_target(tgt, nrm)
        b, e = shard_bounds(len(src), 2, r)
        c.set_source(np.ascontiguousarray(src[b:e]))
        engines.append(HipShardEngine(c, "cuda:0"))
    for eng in engines:
        eng.begin(total_points=len(src))
    for _ in range(5):
        total = torch.zeros(32, dtype=torch.int64, device="cuda:0")
        for eng in engines:
            total += eng.accumulate()
        for eng in engines:
            eng.update(total)
    outs = [eng.finish() for eng in engines]
    # the history's row count comes from the library, also after sharded stepping
    Xs, Is = engines[0].ctx.history()
    assert Xs.shape == (5, 6) and Is.shape == (5, 4)
    assert np.abs(Xs - full.history()[0]).max() < 1e-10
    assert np.array_equal(outs[0][0], outs[1][0])
    assert np.abs(outs[0][0] - T_full).max() < 1e-10
    assert outs[0][1].inliers == info_full.inliers
    moved = np.concatenate([eng.ctx.get_source() for eng in engines])
    assert np.abs(moved - full.get_source()).max() <= 4e-5
    r = po.icp_point2plane(po.KDTree(tgt), nrm, src, 5)
    assert refmath.rot_angle(outs[0][0][:3, :3], r["trans"][:3, :3]) < 1e-6


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import stereovision_amd as sv
    from stereovision_amd import synth
    from stereovision_amd.sharded import HipShardEngine, run_sharded, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    tgt, nrm, src, _ = synth.make_pair(128)
    c = sv.Context(0)
    c.set_target(tgt, nrm)
    b, e = shard_bounds(len(src), world, rank)
    c.set_source(np.ascontiguousarray(src[b:e]))
    T, info = run_sharded(HipShardEngine(c, "cuda:0"), 5)
    np.save(os.path.join(out_dir, f"T{rank}.npy"), T)
    c.close()
    dist.destroy_process_group()


def test_two_processes_share_one_gpu_over_gloo(tmp_path):
    """Two ranks (both on cuda:0, RCCL refuses two ranks per device, so the
    32-double block travels over gloo) against the unsharded run."""
    import torch.multiprocessing as mp
    import stereovision_amd as sv
    from stereovision_amd import synth
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    T0, T1 = np.load(tmp_path / "T0.npy"), np.load(tmp_path / "T1.npy")
    assert np.array_equal(T0, T1)
    tgt, nrm, src, _ = synth.make_pair(128)
    full = sv.Context(0)
    full.set_target(tgt, nrm)
    full.set_source(src)
    T_full, _ = full.run(5)
    assert np.abs(T0 - T_full).max() < 1e-10


def test_calibration_views_split_over_two_contexts(po):
    """Two contexts on one device hold 3 and 4 of 7 views; their 272 sums are added as the
    all-reduce would and handed back."""
    import torch
    from stereovision_amd import synth
    from stereovision_amd.calib import (HipCalEngine, ReprojectionEvaluator, split_views,
                                        view_bounds)
    V = 7
    w, l, r, _, start = synth.make_stereo_views(views=V)
    whole = ReprojectionEvaluator(w, l, r).evaluate(start, want_residual=False)
    engines, spans = [], []
    for rank in range(2):
        b, e = view_bounds(V, 2, rank)
        eng = HipCalEngine(w, l[b:e], r[b:e], "cuda:0")
        eng.set_params(split_views(start, 5, V, True, b, e), 5)
        eng.eval()
        engines.append(eng)
        spans.append((b, e))
    total = torch.zeros(272, dtype=torch.float64, device="cuda:0")
    for eng in engines:
        total += eng.export_sums()
    outs = []
    for eng in engines:
        eng.import_sums(total)
        outs.append(eng.fetch())
    for n in ("LL", "RR", "RS", "SS", "gL", "gR", "gS"):
        assert np.array_equal(outs[0][n], outs[1][n])
        assert np.allclose(outs[0][n], whole[n], rtol=1e-13, atol=1e-13 * np.abs(whole[n]).max())
    assert outs[0]["sq_residual"] == pytest.approx(whole["sq_residual"], rel=1e-13)
    for out, (b, e) in zip(outs, spans):
        for n in ("LE", "EE", "gE"):
            assert np.array_equal(out[n], whole[n][b:e])
    want = po.stereo_eval_sparse(start, 5, w, l, r, False)
    assert np.abs(outs[0]["LL"] - want["LL"]).max() / np.abs(want["LL"]).max() < 1e-6


def test_shard_run_with_one_rank_communicator_equals_run(po):
    """svicp_shard_run (iteration kernel + RCCL all-reduce per iteration, captured in one HIP
    graph) with a one-rank communicator against svicp_run on the same pair, and against the
    oracle; then without a communicator, and eagerly."""
    import stereovision_amd as sv
    from stereovision_amd import synth
    from stereovision_amd.sharded import init_comm, run_sharded_native
    tgt, nrm, src, _ = synth.make_pair(160, pair=2)
    with sv.Context(0) as ref, sv.Context(0) as ctx:
        for c in (ref, ctx):
            c.set_target(tgt, nrm)
            c.set_source(src)
        T_ref, info_ref = ref.run(8)
        X_ref, _ = ref.history()
        T0, info0 = run_sharded_native(ctx, 8)              # no communicator: world size 1
        assert ctx.shard_run_mode() == (True, 1)
        assert np.abs(T0 - T_ref).max() < 1e-10
        init_comm(ctx)
        ctx.set_source(src)
        T1, info1 = run_sharded_native(ctx, 8)
        captured, world = ctx.shard_run_mode()
        assert world == 1
        print("collective captured in the graph:", captured)
        assert np.abs(T1 - T_ref).max() < 1e-10
        assert info1.inliers == info_ref.inliers and info1.iterations == 8
        X1, _ = ctx.history()
        assert np.abs(X1 - X_ref).max() < 1e-10
        assert np.array_equal(ctx.get_source(), ref.get_source())
        T1b, _ = run_sharded_native(ctx, 8)                 # replay
        assert np.abs(T1b - T_ref).max() < 1e-10
        ctx.set_options(use_graph=0)
        T2, _ = run_sharded_native(ctx, 8)
        assert ctx.shard_run_mode()[0] is False
        assert np.abs(T2 - T_ref).max() < 1e-10
        ctx.comm_destroy()
        r = po.icp_point2plane(po.KDTree(tgt), nrm, src, 8)
        assert np.abs(T1 - r["trans"]).max() < 1e-8


def test_config3_pair_registers_through_shard_run():
    """The 16.8 M-point pair of the sharded configuration (16 384 tiles: the first launches run a
    block per tile with the solve in a launch of its own, the later ones a round of blocks that
    work through several tiles and leave unread the tiles whose box records prove that they stay
    put), 30 iterations through svicp_shard_run with a one-rank communicator: it must arrive at
    its generating pose, with nearly every point matched, and a second run must give the same
    bits.  The same bits again with the proofs switched off, and the result of a run that
    searches every point in every iteration (coherent = 0)."""
    import stereovision_amd as sv
    from stereovision_amd import synth
    from stereovision_amd.sharded import init_comm, run_sharded_native
    tgt, nrm, src, P = synth.make_pair(4096, inlier_frac=0.7)
    with sv.Context(0) as ctx:
        ctx.set_target(tgt, nrm)
        ctx.set_source(src)
        init_comm(ctx)
        T, info = run_sharded_native(ctx, 30)
        assert info.status == 0 and info.iterations == 30
        assert np.abs(T @ P - np.eye(4)).max() < 5e-4
        assert info.inliers > 0.98 * len(src)
        counts = ctx.search_counts()
        assert counts[0] == len(src) and counts[-1] < 1000
        unread = ctx.tiles_left_unread()
        tiles = (len(src) + 1023) // 1024
        assert unread[0] == 0 and unread[1] == 0 and unread[-1] > 0.9 * tiles
        X, I = ctx.history()
        moved = ctx.get_source()
        ctx.set_source(src)
        T2, info2 = run_sharded_native(ctx, 30)
        assert np.array_equal(T, T2) and info2.inliers == info.inliers
        ctx.comm_destroy()
        # every tile read in every launch: the same bits
        ctx._check(ctx._lib.svicp_debug_use_tile_boxes(ctx._h, 0))
        ctx.set_source(src)
        T3, info3 = ctx.run(30)
        X3, I3 = ctx.history()
        assert ctx.tiles_left_unread().max() == 0
        assert np.array_equal(T, T3) and np.array_equal(X, X3) and np.array_equal(I, I3)
        assert np.array_equal(moved.view(np.uint32), ctx.get_source().view(np.uint32))
        assert np.array_equal(counts, ctx.search_counts())
        ctx._check(ctx._lib.svicp_debug_use_tile_boxes(ctx._h, 1))
        # and a full search of every point in every iteration
        ctx.set_options(coherent=0)
        ctx.set_source(src)
        T4, info4 = ctx.run(30)
        X4, I4 = ctx.history()
        assert np.abs(X - X4).max() < 1e-10 and np.abs(T - T4).max() < 1e-10
        assert np.array_equal(I[:, 0], I4[:, 0])
        assert ((moved == ctx.get_source()).all(axis=1)).mean() > 0.999
