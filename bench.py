#!/usr/bin/env python3
"""Benchmark of the point-to-plane ICP hot path on MI355X.

A step is one registration of a synthetic 1024^2-point pair with analytic
normals, 30 iterations, clouds resident in HBM (the headline configuration, configs[1]).
With N > 1 ranks every rank registers its own independent pair (no data-path
collective) and `value` is the aggregate iterations/s.  The same run also
times, at every N through the same code path:
  * `value_sharded`: one 4096^2-point pair split over the ranks, the loop inside
    the library (svicp_shard_run: iteration kernel + RCCL all-reduce per
    iteration, captured in one HIP graph) - strong scaling;
  * `batched`: 512 independent 65536-point pairs per rank in one arena batch.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
      --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak
FP64_VECTOR_PEAK_TF = 78.6  # half the 157.3 TF fp32 vector peak: CDNA4 issues fp64 FMAs at half rate


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--side", type=int, default=1024, help="points per cloud = side^2")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--sharded-side", type=int, default=4096,
                    help="side of the pair of the sharded leg; 0 disables the leg")
    ap.add_argument("--cpu-iters", type=int, default=30,
                    help="iterations of the CPU baseline sample; 0 disables it")
    ap.add_argument("--batch-pairs", type=int, default=512,
                    help="pairs (256^2 points each) per rank of the batched leg; 0 disables it")
    ap.add_argument("--extras", type=int, default=1,
                    help="1: also time normal estimation, point-to-point ICP, rectification maps "
                         "and the drop-in call")
    ap.add_argument("--leg-budget-s", type=float, default=480.0,
                    help="seconds the legs after the headline measurement may take in all before "
                         "the line is written without the unfinished ones")
    ap.add_argument("--cal-views", type=int, default=2000,
                    help="views of the calibration leg (11 x 8 board each); 0 disables the leg")
    return ap.parse_args()


def profile_figure(name):
    """A figure kept under profiles/ by the counter passes of this same command (rocprofv3
    --pmc cannot run inside the timed process); None when the file or the key is missing."""
    for f in ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        path = os.path.join(ROOT, "profiles", f)
        if os.path.exists(path):
            with open(path) as fh:
                d = json.load(fh)
            if name in d:
                return d[name], "profiles/" + f
    return None, None


def profile_counters(kernel_contains, regime_contains=""):
    """The derived counter figures (busy fractions, hit rates, wait shares) the counter passes
    left under profiles/ for one kernel and regime; {} when there are none."""
    for f in ("r03_counters.json", "r02_counters.json"):
        path = os.path.join(ROOT, "profiles", f)
        if not os.path.exists(path):
            continue
        with open(path) as fh:
            kernels = json.load(fh).get("kernels", {})
        for name, entry in kernels.items():
            if kernel_contains in name and regime_contains in name:
                out = {k: v for k, v in entry.items() if k != "counters" and not isinstance(v, dict)}
                out["source"] = "profiles/" + f
                return out
    return {}


# Exit status of the process, decided from what the legs left behind: the line is written in
# every case, but a run that gave up on a leg must not look like a clean one to whoever
# records only the status.
EXIT_OK, EXIT_FAILED_LEG, EXIT_UNFINISHED_LEG = 0, 2, 3


def exit_status(failed_legs, unfinished_leg):
    if unfinished_leg:
        return EXIT_UNFINISHED_LEG
    return EXIT_FAILED_LEG if failed_legs else EXIT_OK


def main():
    args = parse()
    # Libraries underneath (RCCL prints a version banner when a communicator is made) write to
    # the process's stdout; the one JSON line must be alone there.  Everything else goes to
    # stderr: descriptor 1 points at stderr while the benchmark runs, and the line is written
    # to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import stereovision_amd as sv
    from stereovision_amd import synth
    from stereovision_amd.batched import HipBatchEngine, pair_bounds, run_pairs_split
    from stereovision_amd.sharded import HipShardEngine, init_comm, run_sharded, run_sharded_native, shard_bounds

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1
    # SVICP_BENCH_BACKEND=gloo rehearses the multi-rank paths where the ranks outnumber the
    # GPUs (ranks then share devices, the exchanges go through the host and the sharded leg
    # falls back to the stepping calls: RCCL refuses two ranks on one device); the default is
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

    def sum_over_ranks(values):
        if not distributed:
            return [float(v) for v in values]
        t = torch.tensor(list(values), dtype=torch.float64, device=f"cuda:{local}" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

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
    searched = ctx.search_counts()                    # points searched per iteration of a step

    # ---- per-launch durations: the same steps, eager, HIP events on the library's stream
    # around every launch (kernel_times' second value is an event pair with nothing between
    # its records: what the pair itself adds to every measured launch) ----
    ctx.set_options(time_kernels=1)
    fused, empty, per_iter = [], [], []
    for _ in range(max(3, min(args.steps, 8))):
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
    # the launches fall in two regimes: those that search (nearly) every point, which do the
    # per-point work the byte model counts, and the settled ones, which skip the tiles whose
    # points no longer move and are bound by latency, not by bytes
    full = searched >= 0.9 * n
    settled = np.zeros(args.iters, bool)
    settled[8:] = searched[8:] < 0.001 * n
    full_ms = float(per_iter[full].mean()) if full.any() else float("nan")
    settled_ms = float(per_iter[settled].mean()) if settled.any() else float("nan")
    achieved_full = bytes_fused / (full_ms * 1e-3) / 1e9
    achieved_blend = bytes_fused / (launch_region_ms * 1e-3) / 1e9
    traffic_full, traffic_src = profile_figure("icp_iterate_full_search_bytes_per_launch")
    traffic_settled, _ = profile_figure("icp_iterate_settled_bytes_per_launch")
    traffic_blend, traffic_blend_src = profile_figure("icp_iterate_kernel_bytes_per_launch")

    nn_capped_ms = ctx.time_nn(exact=False, reps=20)
    nn_exact_ms = ctx.time_nn(exact=True, reps=20)
    loads_ms, loads_instr = ctx.time_gather(True, 0, 20)
    hashed_ms, hashed_instr = ctx.time_gather(False, 16, 5)
    bytes_nn = 20.0 * n + 16.0 * len(tgt)
    # queries/s over all ranks: every rank searches its own million queries in its own replica
    # of a target (queries sharded, target replicated)
    nn_rates = sum_over_ranks([n / (nn_capped_ms * 1e-3), n / (nn_exact_ms * 1e-3),
                               n / (nn_capped_start_ms * 1e-3), n / (nn_exact_start_ms * 1e-3)])
    # without the shortcut for points whose correspondence provably did not change
    ctx.set_options(coherent=0)
    ctx.run(args.iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(2, args.steps // 4)):
        ctx.run(args.iters)
    full_search_value = max(2, args.steps // 4) * args.iters / (time.perf_counter() - t0)
    ctx.set_options(coherent=1)

    # the same pair from the pose that leaves 70 % of the source within the threshold
    _, _, src70, _ = synth.make_pair(args.side, pair=rank, inlier_frac=0.7)
    ctx.set_source(src70)
    ctx.run(args.iters)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(max(2, args.steps // 2)):
        ctx.run(args.iters)
    value_inlier_pose = max(2, args.steps // 2) * args.iters / (time.perf_counter() - t0)
    ctx.set_source(src)
    ctx.run(args.iters)

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

    def build_out():
        # read when the line is written: the legs below fill their variables as they finish
        return {
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
                            "one independent pair per GPU; start pose Rz(0.6 deg) Ry(-0.4 deg) "
                            "Rx(0.5 deg), t = (-0.4, 1.2, 0.8), which leaves 16 % of the source within "
                            "the weight threshold at iteration 0 (value_inlier_pose: the same pair "
                            "from the pose that leaves 70 %)",
                "points": n, "iters_per_step": args.iters,
            },
            # the dominant kernel in the regime the byte model describes: a launch that searches
            # (nearly) every point does the per-point work of the model once per point
            "roofline": {
                "bound": "hbm", "kernel": "icp_iterate_kernel, launches that search >= 90 % of the points",
                "achieved": achieved_full, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_full / HBM_PEAK_GBS,
                "traffic": traffic_full, "traffic_source": traffic_src,
                "bytes_per_launch": bytes_fused, "launch_ms": full_ms, "units_per_launch": n,
                "launch_ms_method": "HIP events on the launch stream around each eager launch, minus an "
                                    "empty event pair; mean over the launches of this regime",
                "launches_of_this_regime_per_step": int(full.sum()),
                "stream_copy_GBs": stream_copy_gbs, "frac_of_stream_copy": achieved_full / stream_copy_gbs,
                "bound_in_practice": "the gather path (see roofline_correspondence.own_loads_ms), not HBM",
                "counters": profile_counters("icp_iterate_kernel", "searches every point"),
            },
            # every launch of the timed region averaged, against the same algorithmic bytes: what
            # the figure called `roofline` was before the regimes were split
            "roofline_effective": {
                "bound": "hbm", "kernel": "icp_iterate_kernel, all launches of the timed region",
                "achieved": achieved_blend, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved_blend / HBM_PEAK_GBS, "traffic": traffic_blend,
                "traffic_source": traffic_blend_src,
                "bytes_per_launch": bytes_fused, "launch_ms": launch_region_ms,
                "launch_ms_method": "HIP events on the launch stream around the whole timed region "
                                    "/ (steps x iterations); gaps between launches included",
                "launches_averaged": args.steps * args.iters,
                "note": "most launches skip the tiles whose points no longer move: not a bandwidth fraction",
            },
            # a settled launch: about 900 of 1024 tiles exit after reading their coordinates
            "roofline_settled": {
                "bound": "latency", "kernel": "icp_iterate_kernel, launches that search < 0.1 % of the points",
                "launch_ms": settled_ms, "traffic": traffic_settled, "traffic_source": traffic_src,
                "achieved": (traffic_settled or 0.0) / (settled_ms * 1e-3) / 1e9 if traffic_settled else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ((traffic_settled / (settled_ms * 1e-3) / 1e9) / HBM_PEAK_GBS) if traffic_settled else None,
                "launches_of_this_regime_per_step": int(settled.sum()),
            },
            # the standalone correspondence kernel (the reduction has no kernel of its own: it is
            # fused into icp_iterate_kernel)
            "roofline_correspondence": {
                "bound": "hbm", "kernel": "nn1_kernel<false>", "unit": "GB/s", "peak": HBM_PEAK_GBS,
                "achieved": bytes_nn / (nn_capped_ms * 1e-3) / 1e9,
                "frac": bytes_nn / (nn_capped_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "bytes_per_launch": bytes_nn, "launch_ms": nn_capped_ms,
                # the kernel's vector-memory instructions alone (same look-ups, same candidate
                # loads, no arithmetic, nothing written): the floor of a per-lane gather search
                "own_loads_ms": loads_ms, "own_loads_wave_instructions": loads_instr,
                "frac_of_own_loads": loads_ms / nn_capped_ms,
                "gather_path_wave_instructions_per_s": loads_instr / (loads_ms * 1e-3),
                "hashed_loads_wave_instructions_per_s": hashed_instr / (hashed_ms * 1e-3),
                "note": "20 B per query + 16 B per target point; bound by the texture-address path "
                        "(one 16-byte gather per candidate per lane), not by HBM: the search runs at "
                        "frac_of_own_loads of the speed of its own loads",
                "counters": profile_counters("nn1_kernel<false"),
                "own_loads_counters": profile_counters("gather_probe_kernel"),
            },
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_all,
            "kernels_ms": {"icp_iterate": fused_ms,
                           "icp_iterate_per_iteration": [round(float(x), 5) for x in per_iter],
                           "searched_points_per_iteration": [int(x) for x in searched],
                           "nn1_capped": nn_capped_ms, "nn1_exact": nn_exact_ms,
                           "eager_launch_ms_between_events": raw_ms, "empty_event_pair_ms": pair_ms},
            "value_searching_every_point_every_iteration": full_search_value,
            "value_inlier_pose": value_inlier_pose,
            "value_sharded": value_sharded,
            "nn_queries_per_s": {"capped": nn_rates[0],
                                 "exact": nn_rates[1],
                                 "capped_GBs": bytes_nn / (nn_capped_ms * 1e-3) / 1e9,
                                 "positions": "after registration (every source point has a close "
                                              "target point)",
                                 "capped_at_start": nn_rates[2],
                                 "exact_at_start": nn_rates[3],
                                 "summed_over_ranks": world,
                                 "note": "whole-job rates: each rank's own queries against its own "
                                         "replica of a target; capped_GBs is rank 0's"},
            "grid_build_ms": grid_build_s * 1e3,
            "grid_build_ms_buffers_in_place": grid_rebuild_s * 1e3,
            "final_step_norm": info.step_norm,
            "sharded": sharded,
            "batched": batched,
            "calibration": calibration,
            "dropin": dropin,
            "neighbouring_rows": neighbours,
        }

    # ---- everything below is secondary to `value`.  A leg that raises leaves its error in the
    # line instead of taking the headline with it; a leg that does not come back (a collective
    # that never completes on a machine this code has not met) is cut off by a watchdog, which
    # writes the line with what has finished and ends the process.
    sharded = value_sharded = batched = calibration = neighbours = dropin = cpu = cpu_all = None
    failed = {}
    current = ["headline"]
    written = threading.Lock()
    finished = threading.Event()

    def write_line(unfinished=None):
        if not written.acquire(blocking=False):
            return
        if rank == 0:
            out = build_out()
            if failed:
                out["failed_legs"] = failed
            if unfinished:
                out["unfinished_leg"] = unfinished
            sys.stdout.flush()
            os.write(json_fd, (json.dumps(out) + "\n").encode())

    def watchdog():
        if not finished.wait(args.leg_budget_s):
            unfinished = f"{current[0]}: no return within {args.leg_budget_s} s of the headline measurement"
            # every rank names the leg it is stuck in; rank 0 writes the line with what has finished
            os.write(2, f"bench.py rank {rank}: giving up on {unfinished}\n".encode())
            write_line(unfinished)
            os._exit(exit_status(failed, unfinished))
    threading.Thread(target=watchdog, daemon=True).start()

    def guard(name, leg):
        current[0] = name
        try:
            return leg()
        except Exception as e:                      # noqa: BLE001 - reported in the line
            failed[name] = f"{type(e).__name__}: {e}"[:400]
            os.write(2, f"bench.py rank {rank}: leg {name} failed: {failed[name]}\n".encode())
            return None

    # ---- sharded leg: one large pair, source split over the ranks, the loop inside the
    # library at every N (N = 1 included: a one-rank communicator, the same graph) ----
    def leg_sharded():
        nonlocal value_sharded
        st, sn, ss, spose = synth.make_pair(args.sharded_side, pair=0, inlier_frac=0.7)
        b0, e0 = shard_bounds(len(ss), world, rank)
        sctx = sv.Context(local)
        sctx.set_target(st, sn)
        sctx.set_source(np.ascontiguousarray(ss[b0:e0]))
        n_sh = len(ss)
        del ss
        native = backend == "nccl"
        if native:
            init_comm(sctx)
            run_once = lambda: run_sharded_native(sctx, args.iters)
        else:
            eng = HipShardEngine(sctx, f"cuda:{local}")
            run_once = lambda: run_sharded(eng, args.iters)
        Ts, sinfo = run_once()
        barrier()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            Ts, sinfo = run_once()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        value_sharded = reps * args.iters / dt
        captured, comm_world = sctx.shard_run_mode() if native else (False, world)
        sharded = {
            "workload": f"configs[2]: one {args.sharded_side}^2-point pair (start pose with 70 % of the "
                        f"source within the threshold), source split over {world} rank(s), target "
                        "replicated, one all-reduce of 128 64-bit fixed-point sums per iteration",
            "iters_per_s": value_sharded, "scaling": "strong", "points": n_sh,
            "path": "svicp_shard_run (iteration kernel + ncclAllReduce per iteration on one stream)"
                    if native else "stepping calls + torch.distributed (rehearsal backend)",
            "graph_captured": bool(captured), "communicator_world": comm_world,
            "pose_error": float(np.abs(Ts @ spose - np.eye(4)).max()),
            "measured_on": "this run's GPUs; N > 1 has not been measured on hardware by the builder",
        }
        if native:
            sctx.comm_destroy()
        sctx.close()
        return sharded
    if args.sharded_side > 0:
        sharded = guard("sharded", leg_sharded)

    # ---- batched leg: configs[3]'s per-GPU slice, 512 independent 65536-point pairs per
    # rank in one arena batch (one launch per iteration for all); ranks take disjoint pairs,
    # no collective while iterating, one gather of the transforms at the end ----
    def leg_batched():
        total_pairs = args.batch_pairs * world
        first, last = pair_bounds(total_pairs, world, rank)
        tg, nr, sr = [], [], []
        for pid in range(first, last):
            bt, bn, bs, _ = synth.make_pair(256, pair=pid, pose=synth.small_pose(pid))
            tg.append(bt); nr.append(bn); sr.append(bs)
        eng = HipBatchEngine(tg, nr, sr, device=local)
        del tg, nr, sr
        _, setup_ms, arena_bytes = eng.batch.info()
        for _ in range(3):
            eng.run(args.iters)
        barrier()
        t0 = time.perf_counter()
        reps = 10
        run_ms, dev_ms = [], []
        for _ in range(reps):
            t1 = time.perf_counter()
            eng.run(args.iters)
            run_ms.append((time.perf_counter() - t1) * 1e3)
            dev_ms.append(eng.batch.last_ms())
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        t0 = time.perf_counter()
        Tall, _ = run_pairs_split(eng, total_pairs, args.iters)
        gather_s = time.perf_counter() - t0
        batched = {
            "workload": f"configs[3] slice: {args.batch_pairs} independent pairs of 65536 points per rank "
                        f"({total_pairs} in all), {args.iters} iterations, per-pair pose, one launch per "
                        "iteration for all pairs of a rank",
            "pair_iterations_per_s": world * reps * args.batch_pairs * args.iters / dt,
            "point_iterations_per_s": world * reps * args.batch_pairs * args.iters * 65536 / dt,
            "device_ms_per_run": eng.batch.last_ms(),
            "host_ms_per_run": dt / reps * 1e3,
            "host_ms_runs": [round(x, 2) for x in run_ms], "device_ms_runs": [round(x, 2) for x in dev_ms],
            "setup_ms": setup_ms, "arena_GiB": arena_bytes / 2 ** 30,
            "run_and_gather_ms": gather_s * 1e3,
            "scaling": "weak",
        }
        eng.close()
        return batched
    if args.batch_pairs > 0:
        batched = guard("batched", leg_batched)

    # ---- calibration leg: residual + Jacobian evaluations of the stereo refinement,
    # views split over the ranks, one 272-double all-reduce per evaluation ----
    def leg_calibration():
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
        C = cl.shape[1]
        cal_bytes = 56.0 * V * C + 48.0 * V
        # fp64 operations per (view, corner, camera), five distortion terms, the least the
        # central differences need: 13 geometric stages (12 multiply-adds of the 3 x 4 product,
        # two quotients, r: 26 operations), 31 pixel stages (distortion polynomial, tangential
        # terms, intrinsics: 36), 30 central differences (a subtraction and a quotient each), the
        # residual, and two rank-1 updates of the SYMMETRIC 16 x 16 Gram matrix (the upper
        # triangle: 136 multiply-adds = 272 operations each; the matrix instruction computes
        # all 256 entries, the other 120 are not counted); an operation is one add, multiply
        # or divide.  (The kernel deals one component of a corner's rows to a lane and spends 32
        # stages of each kind.)
        flops_unit = 13 * 26 + 31 * 36 + 30 * 2 + 2 + 2 * 272
        flops_eval = flops_unit * 2.0 * V * C
        tf = flops_eval / (eval_ms * 1e-3) / 1e12
        calibration = {
            "workload": f"configs[4]: stereo reprojection residual + central-difference Jacobian reduced to "
                        f"block-sparse normal equations, {V} views x {C} corners "
                        f"({4 * V * C} residual rows, {18 + 6 * V + 6} parameters), "
                        f"views split over {world} rank(s)",
            "evaluations_per_s": 1e3 / eval_ms, "ms_per_evaluation": eval_ms, "dtype": "f64",
            "algorithmic_GBs": cal_bytes / (eval_ms * 1e-3) / 1e9,
            "scaling": "strong",
            "roofline": {
                "bound": "fp64 vector", "kernel": "cal_eval_kernel<5> + cal_reduce_kernel",
                "achieved": tf, "peak": FP64_VECTOR_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_VECTOR_PEAK_TF,
                "flops_per_view_corner_camera": flops_unit, "units": 2 * V * C,
                "note": "the loop issues 124 vector instructions (102 fp64) and 2 matrix instructions per "
                        "four corners at 6.6 SIMD cycles per wave instruction, whatever the occupancy; "
                        "HBM is not the bound (algorithmic_GBs)",
                "peak_source": "half the fp32 vector peak (157.3 TF) of MI355X_MICROARCH.md",
            },
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
                          f"{4 * V * C} x {18 + 6 * V + 6} doubles cannot be allocated)",
            }
        return calibration
    if args.cal_views >= world:
        calibration = guard("calibration", leg_calibration)

    # ---- the rows next to the hot path, rank 0 only, small and bounded: normal estimation,
    # the point-to-point variant, the rectification maps, the drop-in call from host buffers;
    # each with the CPU checker beside it
    def leg_extras():
        nonlocal dropin
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
        # the drop-in call as CPoint2PlaneICP::run of the compat classes makes it
        # (CPoint2PlaneICP.cpp:31-62): host buffer in, 30 iterations, moved cloud back out
        from stereovision_amd.icp import pinned_empty

        def call(buf_in, buf_out):
            t0 = time.perf_counter()
            ectx.set_source(buf_in)
            ectx.run(args.iters)
            ectx.get_source(buf_out)
            return (time.perf_counter() - t0) * 1e3
        out_pageable = np.empty_like(src)
        pin_in, pin_out = pinned_empty(src.shape), pinned_empty(src.shape)
        pin_in[:] = src
        call(src, out_pageable); call(pin_in, pin_out)
        dropin = {
            "workload": f"set_source + {args.iters} iterations + get_source, {n} points, host buffers "
                        "(PCIe inclusive; never `value`)",
            "dropin_call_ms": float(np.mean([call(src, out_pageable) for _ in range(5)])),
            "dropin_call_ms_pinned_buffers": float(np.mean([call(pin_in, pin_out) for _ in range(5)])),
            "note": "pinned: the clouds live in svicp_host_alloc memory, as the drop-in CTxtReader allocates them",
        }
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
        cr_, sr_ = np.cos(0.01), np.sin(0.01)
        Rrig = np.array([[cr_, 0, sr_], [0, 1, 0], [-sr_, 0, cr_]])
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
            po.icp_point2point(ptree, src, 5, faithful=True, threads=1, history=False)
            neighbours["point_to_point"]["cpu_baseline"] = {
                "value": 5 / (time.perf_counter() - t0), "unit": "iters/s", "cores": 1, "kind": "port",
                "sample": "5 iterations of the same pair, tree build excluded"}
            t0 = time.perf_counter()
            po.rectify_maps(1920, 1080, KL, dL, KR, dR, pl, pr)
            neighbours["rectify_maps"]["cpu_baseline"] = {
                "value": (time.perf_counter() - t0) * 1e3, "unit": "ms", "cores": 1, "kind": "port",
                "sample": "the same 1920 x 1080 maps (the inverse projection formed once, not per "
                          "pixel as the reference does)"}
        return neighbours
    if rank == 0 and args.extras:
        neighbours = guard("neighbouring_rows", leg_extras)

    # ---- CPU baseline: the oracle on the same pair, one thread ------------
    current[0] = "cpu_baseline"
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

    if cpu is not None:
        threads = min(po.max_threads(), 16)
        t0 = time.perf_counter()
        po.icp_point2plane(tree, nrm, src, args.cpu_iters, faithful=True, threads=threads, history=False)
        cpu_all = {"value": args.cpu_iters / (time.perf_counter() - t0), "unit": "iters/s",
                   "cores": threads, "kind": "port",
                   "sample": "as cpu_baseline, the nearest-neighbour queries spread over the host "
                             "threads (the reference itself is single-threaded)"}

    finished.set()
    write_line()
    ctx.close()
    if distributed:
        dist.destroy_process_group()
    # the line is out; a leg that failed still makes the run a failed one
    status = exit_status(failed, None)
    if status != EXIT_OK:
        sys.exit(status)


if __name__ == "__main__":
    main()
