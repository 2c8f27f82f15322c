#!/usr/bin/env python3
"""Benchmark of the point-to-plane ICP hot path on MI355X.

A step is one registration of a synthetic 1024^2-point pair with analytic
normals, 30 iterations, clouds resident in HBM (BASELINE.json configs[1]).
With N > 1 ranks every rank registers its own independent pair (batch mode,
no data-path collective) and `value` is the aggregate iterations/s; the same
run also times the sharded mode (one 4096^2-point pair split over the ranks,
one 32-double all-reduce per iteration) and reports it under "sharded".

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--side", type=int, default=1024, help="points per cloud = side^2")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--sharded-side", type=int, default=4096,
                    help="side of the pair of the sharded leg; 0 disables the leg")
    ap.add_argument("--cpu-iters", type=int, default=30,
                    help="iterations of the CPU baseline sample; 0 disables it")
    ap.add_argument("--batch-pairs", type=int, default=64,
                    help="pairs (256^2 points each) per rank of the batched leg; 0 disables it")
    ap.add_argument("--extras", type=int, default=1,
                    help="1: also time normal estimation, point-to-point ICP and rectification maps")
    ap.add_argument("--cal-views", type=int, default=2000,
                    help="views of the calibration leg (11 x 8 board each); 0 disables the leg")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    import stereovision_amd as sv
    from stereovision_amd import synth
    from stereovision_amd.sharded import HipShardEngine, run_sharded, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # SVICP_BENCH_BACKEND=gloo rehearses the multi-rank paths where the ranks outnumber the
    # GPUs (ranks then share devices and the exchanges go through the host); the default is
    # RCCL, one rank per GPU
    backend = os.environ.get("SVICP_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    def barrier():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n = args.side * args.side
    tgt, nrm, src, pose = synth.make_pair(args.side, pair=rank)
    ctx = sv.Context(local)
    t0 = time.perf_counter()
    ctx.set_target(tgt, nrm)
    grid_build_s = time.perf_counter() - t0           # first call: device allocations included
    t0 = time.perf_counter()
    ctx.set_target(tgt, nrm)
    grid_rebuild_s = time.perf_counter() - t0         # the same with the buffers in place
    ctx.set_source(src)

    # nearest-neighbour kernels at the starting positions (16 % of the source has a target
    # point within the threshold there; the exact search has to look far for the rest)
    nn_capped_start_ms = ctx.time_nn(exact=False, reps=20)
    nn_exact_start_ms = ctx.time_nn(exact=True, reps=5)

    # ---- primary: K registrations of `iters` iterations each -------------
    # The library launches on a torch stream here so that torch's events, recorded on that
    # same stream, bracket exactly the launches of the timed region.
    side = torch.cuda.Stream(device=f"cuda:{local}")
    ctx.set_stream(side.cuda_stream)
    for _ in range(args.warmup):
        T, info = ctx.run(args.iters)
    barrier()
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record(side)
    for _ in range(args.steps):
        T, info = ctx.run(args.iters)
    ev_b.record(side)
    barrier()
    elapsed = time.perf_counter() - t0
    region_ms = ev_a.elapsed_time(ev_b)
    launch_region_ms = region_ms / (args.steps * args.iters)
    elapsed = max_over_ranks(elapsed)
    value = world * args.steps * args.iters / elapsed

    # ---- per-kernel durations: the same steps, eager, HIP events on the
    # library's stream around every launch -------------------------------
    # (the second value kernel_times returns is an event pair with nothing between
    # its two records: what the pair itself adds to every measured launch)
    ctx.set_options(time_kernels=1)
    fused, empty, per_iter = [], [], []
    for _ in range(max(2, min(args.steps, 5))):
        ctx.run(args.iters)
        f, s, _ = ctx.kernel_times()
        fused.append(f)
        empty.append(s)
        per_iter.append(ctx.iteration_times()[0])
    ctx.set_options(time_kernels=0)
    raw_ms, pair_ms = float(np.mean(fused)), float(np.mean(empty))
    fused_ms = raw_ms - pair_ms
    per_iter = np.mean(np.stack(per_iter), axis=0) - pair_ms
    # algorithmic bytes of one launch of the iteration kernel: source read + write
    # (24 B/pt), target xyz + index + normal (28 B/pt), each touched once
    bytes_fused = 24.0 * n + 28.0 * len(tgt)
    # the roofline figure uses the average over the timed region (events on the launch
    # stream around all K x iters launches: gaps between launches and the one small solve
    # launch that ends a step included); the per-launch event figures are kept beside it
    achieved = bytes_fused / (launch_region_ms * 1e-3) / 1e9
    slowest_ms = float(per_iter.max())            # a launch that searches every point
    nn_capped_ms = ctx.time_nn(exact=False, reps=20)
    nn_exact_ms = ctx.time_nn(exact=True, reps=20)
    bytes_nn = 20.0 * n + 16.0 * len(tgt)
    # without the shortcut for points whose correspondence provably did not change
    ctx.set_options(coherent=0)
    ctx.run(args.iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(2, args.steps // 2)):
        ctx.run(args.iters)
    full_search_value = max(2, args.steps // 2) * args.iters / (time.perf_counter() - t0)
    ctx.set_options(coherent=1)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
    if os.path.exists(tpath):
        with open(tpath) as fh:
            traffic = json.load(fh).get("icp_iterate_kernel_bytes_per_launch")

    # ---- what this GPU's HBM delivers on a plain copy, for scale (torch is plumbing here:
    # one large device-to-device copy, read + write counted) ----
    a = torch.empty(1 << 28, dtype=torch.uint8, device=f"cuda:{local}")
    b = torch.empty_like(a)
    b.copy_(a)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(10):
        b.copy_(a)
    ev1.record()
    torch.cuda.synchronize()
    stream_copy_gbs = 10 * 2.0 * a.numel() / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
    del a, b

    # ---- sharded leg ----------------------------------------------------
    sharded = None
    if args.sharded_side > 0:
        st, sn, ss, _ = synth.make_pair(args.sharded_side, pair=0)
        b, e = shard_bounds(len(ss), world, rank)
        sctx = sv.Context(local)
        sctx.set_target(st, sn)
        sctx.set_source(np.ascontiguousarray(ss[b:e]))
        del ss
        if distributed:
            eng = HipShardEngine(sctx, f"cuda:{local}")
            run_sharded(eng, args.iters)
            barrier()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                Ts, _ = run_sharded(eng, args.iters)
            barrier()
            dt = time.perf_counter() - t0
            dt = max_over_ranks(dt)
        else:
            sctx.run(args.iters)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = 3
            for _ in range(reps):
                Ts, _ = sctx.run(args.iters)
            dt = time.perf_counter() - t0
        sharded = {
            "workload": f"one {args.sharded_side}^2-point pair, source split over {world} rank(s), "
                        "target replicated, 32-double all-reduce per iteration",
            "iters_per_s": reps * args.iters / dt,
            "scaling": "strong",
        }
        sctx.close()

    # ---- batched leg: many small independent pairs per rank, one launch per
    # iteration for all of them; ranks take disjoint pairs, no collective ----
    batched = None
    if args.batch_pairs > 0:
        bctx = []
        for i in range(args.batch_pairs):
            pid = rank * args.batch_pairs + i
            bt, bn, bs, _ = synth.make_pair(256, pair=pid, pose=synth.small_pose(pid))
            c = sv.Context(local)
            c.set_target(bt, bn)
            c.set_source(bs)
            bctx.append(c)
        batch = sv.Batch(bctx)
        batch.run(args.iters)
        barrier()
        t0 = time.perf_counter()
        reps = 20
        for _ in range(reps):
            batch.run(args.iters)
        barrier()
        dt = time.perf_counter() - t0
        dt = max_over_ranks(dt)
        batched = {
            "workload": f"{args.batch_pairs} independent pairs of 65536 points per rank, "
                        f"{args.iters} iterations, per-pair pose, one launch per iteration",
            "pair_iterations_per_s": world * reps * args.batch_pairs * args.iters / dt,
            "point_iterations_per_s": world * reps * args.batch_pairs * args.iters * 65536 / dt,
            "device_ms_per_run": batch.last_ms(),
            "scaling": "weak",
        }
        batch.close()
        for c in bctx:
            c.close()

    # ---- calibration leg: residual + Jacobian evaluations of the stereo refinement,
    # views split over the ranks, one 272-double all-reduce per evaluation ----
    calibration = None
    if args.cal_views >= world:
        from stereovision_amd.calib import HipCalEngine, evaluate_sharded, view_bounds
        V = args.cal_views
        cw, cl, cr, _, cstart = synth.make_stereo_views(views=V)
        vb, ve = view_bounds(V, world, rank)
        eng = HipCalEngine(cw, cl[vb:ve], cr[vb:ve], f"cuda:{local}")
        reps = 200
        if distributed:
            for _ in range(5):
                evaluate_sharded(eng, cstart, 5, V, vb, ve, fetch=False)
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                evaluate_sharded(eng, cstart, 5, V, vb, ve, fetch=False)
            barrier()
            dt = time.perf_counter() - t0
            eval_ms = max_over_ranks(dt) / reps * 1e3
        else:
            eng.set_params(cstart, 5)
            eval_ms = eng.ev.time_eval(reps)      # HIP events around back-to-back evaluations
        cal_bytes = 56.0 * V * cl.shape[1] + 48.0 * V
        calibration = {
            "workload": f"stereo reprojection residual + central-difference Jacobian reduced to "
                        f"block-sparse normal equations, {V} views x {cl.shape[1]} corners "
                        f"({4 * V * cl.shape[1]} residual rows, {18 + 6 * V + 6} parameters), "
                        f"views split over {world} rank(s)",
            "evaluations_per_s": 1e3 / eval_ms, "ms_per_evaluation": eval_ms, "dtype": "f64",
            "algorithmic_GBs": cal_bytes / (eval_ms * 1e-3) / 1e9,
            "scaling": "strong",
        }
        if not distributed:
            # the caller of the evaluation: one damped step of the refinement loop on the device
            # (per-view blocks eliminated first), step and predicted decrease copied back
            eng.ev.eval()
            eng.ev.solve_damped(1e-3)
            t0 = time.perf_counter()
            for _ in range(20):
                eng.ev.solve_damped(1e-3)
            calibration["damped_step_ms"] = (time.perf_counter() - t0) / 20 * 1e3
        else:
            # the same step with the views split: every rank eliminates its own views, the
            # shared system is summed over the ranks and solved by each, the step is gathered
            from stereovision_amd.calib import solve_damped_sharded
            evaluate_sharded(eng, cstart, 5, V, vb, ve, fetch=False)
            solve_damped_sharded(eng, 1e-3, 5, V, vb, ve)
            barrier()
            t0 = time.perf_counter()
            for _ in range(20):
                solve_damped_sharded(eng, 1e-3, 5, V, vb, ve)
            barrier()
            calibration["damped_step_ms"] = max_over_ranks(time.perf_counter() - t0) / 20 * 1e3
        if rank == 0 and world == 1 and args.cpu_iters > 0:
            from oracle import pyoracle as po
            t0 = time.perf_counter()
            nrep = 0
            while time.perf_counter() - t0 < 5.0:
                po.stereo_eval_sparse(cstart, 5, cw, cl, cr, want_residual=False)
                nrep += 1
            calibration["cpu_baseline"] = {
                "value": nrep / (time.perf_counter() - t0), "unit": "evaluations/s", "cores": 1,
                "kind": "port",
                "sample": f"{nrep} block-sparse evaluations of the same {V}-view problem "
                          "(oracle/cal_oracle.c; the reference's dense Jacobian of "
                          f"{4 * V * cl.shape[1]} x {18 + 6 * V + 6} doubles cannot be allocated)",
            }

    # ---- the rows next to the hot path, rank 0 only, small and bounded: normal estimation,
    # the point-to-point variant, the rectification maps; each with the CPU checker beside it
    neighbours = None
    if rank == 0 and args.extras:
        neighbours = {}
        ectx = sv.Context(local)
        ectx.set_target(tgt, None)                              # allocations
        t0 = time.perf_counter()
        ectx.set_target(tgt, None)                              # grid + 20-NN + eigenvectors
        t_est = time.perf_counter() - t0
        t0 = time.perf_counter()
        ectx.set_target(tgt, nrm)
        t_given = time.perf_counter() - t0
        neighbours["normal_estimation"] = {
            "workload": f"calcNormals on {len(tgt)} target points (20-NN, uncentred moments, "
                        "smallest eigenvector): the whole set_target host call, upload and grid "
                        "build included (the CPU figure includes its tree build)",
            "ms": t_est * 1e3, "set_target_with_given_normals_ms": t_given * 1e3,
            "note": "the second call uploads 12 B of normals per point instead of estimating them"}
        ectx.close()
        pctx = sv.Context(local, dist_threshold=15.0)
        pctx.set_target(tgt, tgt)
        pctx.set_source(src)
        pctx.run_point2point(args.iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            Tp, _ = pctx.run_point2point(args.iters)
        dtp = (time.perf_counter() - t0) / 5
        neighbours["point_to_point"] = {
            "workload": f"CPoint2PointICP, same {n}-point pair, threshold 15, {args.iters} iterations",
            "iters_per_s": args.iters / dtp}
        pctx.close()
        from stereovision_amd.calib import rectify_maps, stereo_rectify
        KL = np.array([[1400.0, 0, 960.0], [0, 1400.0, 540.0], [0, 0, 1]])
        KR = np.array([[1395.0, 0, 955.0], [0, 1398.0, 545.0], [0, 0, 1]])
        dL = np.array([-0.28, 0.09, 1e-3, -5e-4, 0.01])
        dR = np.array([-0.27, 0.08, -7e-4, 4e-4, -0.02])
        cr, sr = np.cos(0.01), np.sin(0.01)
        Rrig = np.array([[cr, 0, sr], [0, 1, 0], [-sr, 0, cr]])
        pl, pr, _ = stereo_rectify(KL, KR, Rrig, [-90.0, 1.0, 2.0])
        rectify_maps(1920, 1080, KL, dL, KR, dR, pl, pr, device=local)
        *_, rect_ms = rectify_maps(1920, 1080, KL, dL, KR, dR, pl, pr, device=local, return_ms=True)
        neighbours["rectify_maps"] = {
            "workload": "doImageRectify, 1920 x 1080, both cameras, five distortion terms",
            "kernel_ms": rect_ms, "GBs": 1920 * 1080 * 32 / (rect_ms * 1e-3) / 1e9}
        if world == 1 and args.cpu_iters > 0:
            from oracle import pyoracle as po
            sub = 1 << 17                                       # a bounded sample for the CPU
            t0 = time.perf_counter()
            po.calc_normals(po.KDTree(tgt[:sub]), threads=1)
            neighbours["normal_estimation"]["cpu_baseline"] = {
                "value": sub / (time.perf_counter() - t0), "unit": "points/s", "cores": 1, "kind": "port",
                "sample": f"first {sub} target points (tree build included)",
                "gpu_points_per_s": len(tgt) / t_est}
            ptree = po.KDTree(tgt)
            t0 = time.perf_counter()
            rp = po.icp_point2point(ptree, src, 5, faithful=True, threads=1, history=False)
            neighbours["point_to_point"]["cpu_baseline"] = {
                "value": 5 / (time.perf_counter() - t0), "unit": "iters/s", "cores": 1, "kind": "port",
                "sample": "5 iterations of the same pair, tree build excluded"}
            t0 = time.perf_counter()
            po.rectify_maps(1920, 1080, KL, dL, KR, dR, pl, pr)
            neighbours["rectify_maps"]["cpu_baseline"] = {
                "value": (time.perf_counter() - t0) * 1e3, "unit": "ms", "cores": 1, "kind": "port",
                "sample": "the same 1920 x 1080 maps (the inverse projection formed once, not per "
                          "pixel as the reference does)"}

    # ---- CPU baseline: the oracle on the same pair, one thread ------------
    cpu = None
    if rank == 0 and world == 1 and args.cpu_iters > 0:
        from oracle import pyoracle as po
        tree = po.KDTree(tgt)                       # index build excluded, as test.cpp:20-21
        t0 = time.perf_counter()
        r = po.icp_point2plane(tree, nrm, src, args.cpu_iters, faithful=True, threads=1,
                               history=False)
        dt = time.perf_counter() - t0
        Xg, _ = ctx.history()
        cpu = {
            "value": args.cpu_iters / dt, "unit": "iters/s", "cores": 1, "kind": "port",
            "sample": f"{args.cpu_iters} of {args.iters} iterations of the same "
                      f"{n}-point pair, kd-tree build excluded, restated reference "
                      "(oracle/icp_oracle.c; the reference needs FLANN+Eigen and cannot be built)",
            "host_cpus": os.cpu_count(),
            "max_abs_X_diff_vs_gpu": float(np.abs(Xg[: args.cpu_iters] - r["X"]).max()),
        }

    cpu_all = None
    if cpu is not None:
        threads = min(po.max_threads(), 16)
        t0 = time.perf_counter()
        po.icp_point2plane(tree, nrm, src, args.cpu_iters, faithful=True, threads=threads, history=False)
        cpu_all = {"value": args.cpu_iters / (time.perf_counter() - t0), "unit": "iters/s",
                   "cores": threads, "kind": "port",
                   "sample": "as cpu_baseline, the nearest-neighbour queries spread over the host "
                             "threads (the reference itself is single-threaded)"}

    if rank == 0:
        out = {
            "metric": "point-to-plane ICP iters/s at 1M pts",
            "value": value,
            "unit": "iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 points and rows, f64 accumulate and solve",
            "data": "synthetic",
            "config": {
                "workload": f"configs[1]: synthetic {args.side}^2-point source/target pair with "
                            f"normals, point-to-plane ICP, {args.iters} iters per step, "
                            "one independent pair per GPU",
                "points": n, "iters_per_step": args.iters,
            },
            "roofline": {
                "bound": "hbm", "kernel": "icp_iterate_kernel",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "bytes_per_launch": bytes_fused, "launch_ms": launch_region_ms,
                "launch_ms_method": "HIP events on the launch stream around the whole timed region "
                                    "/ (steps x iterations)",
                "eager_launch_ms_between_events": raw_ms, "empty_event_pair_ms": pair_ms,
                "eager_launch_ms_minus_empty_pair": fused_ms,
                "launches_averaged": args.steps * args.iters,
                "stream_copy_GBs": stream_copy_gbs, "frac_of_stream_copy": achieved / stream_copy_gbs,
                "loop_ms_per_iteration": elapsed / args.steps / args.iters * 1e3,
                "slowest_launch_ms": slowest_ms,
                "frac_slowest_launch": bytes_fused / (slowest_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            # the standalone correspondence kernel (the reduction has no kernel of its own: it is
            # fused into icp_iterate_kernel, whose roofline is the object above)
            "roofline_correspondence": {
                "bound": "hbm", "kernel": "nn1_kernel<false>", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": bytes_nn / (nn_capped_ms * 1e-3) / 1e9,
                "frac": bytes_nn / (nn_capped_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "bytes_per_launch": bytes_nn, "launch_ms": nn_capped_ms,
                "note": "20 B per query + 16 B per target point; the kernel is bound by its gather "
                        "instructions (texture-address units), not by HBM",
            },
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "kernels_ms": {"icp_iterate": fused_ms,
                           "icp_iterate_per_iteration": [round(float(x), 5) for x in per_iter],
                           "nn1_capped": nn_capped_ms, "nn1_exact": nn_exact_ms},
            "value_searching_every_point_every_iteration": full_search_value,
            "nn_queries_per_s": {"capped": n / (nn_capped_ms * 1e-3),
                                 "exact": n / (nn_exact_ms * 1e-3),
                                 "capped_GBs": bytes_nn / (nn_capped_ms * 1e-3) / 1e9,
                                 "positions": "after registration (every source point has a close "
                                              "target point)",
                                 "capped_at_start": n / (nn_capped_start_ms * 1e-3),
                                 "exact_at_start": n / (nn_exact_start_ms * 1e-3)},
            "grid_build_ms": grid_build_s * 1e3,
            "grid_build_ms_buffers_in_place": grid_rebuild_s * 1e3,
            "final_step_norm": info.step_norm,
            "sharded": sharded,
            "batched": batched,
            "calibration": calibration,
            "neighbouring_rows": neighbours,
        }
        print(json.dumps(out))
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
