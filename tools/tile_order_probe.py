"""Development probe: does an order of the tiles by their cost even out the launches that search
every point?  Mode "costs" (library built with -DSVICP_TRACE): every tile's own time in
iterations 0..2 -> gpurun_out/tile_costs.npy.  Mode "measure" (the product library): the times of
the first launches with the built-in order and with the tiles of each XCD dealt longest first over
its compute units in a snake (svicp_debug_set_tile_order), by the costs of iteration 1."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stereovision_amd as sv
from stereovision_amd import synth, _lib
import ctypes as C

tgt, nrm, src, P = synth.make_pair(1024)
mode = sys.argv[1]
if mode == "costs":
    ctx = sv.Context(0)
    ctx.set_target(tgt, nrm); ctx.set_source(src)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    ctx.run(4); lib.svicp_debug_trace_clear(); ctx.run(4)
    NI, NT, NS = 64, 1024, 16
    buf = np.zeros(NI * NT * NS, dtype=np.uint64)
    lib.svicp_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    assert lib.svicp_debug_trace(buf.ctypes.data, buf.size) == 0
    tr = buf.reshape(NI, NT, NS)[:, :, :8].astype(np.float64) * 0.01
    costs = np.stack([np.where(tr[k] > 0, tr[k], 0).max(axis=1) - tr[k][:, 0] for k in range(3)])
    np.save(os.path.join(ROOT, "gpurun_out", "tile_costs.npy"), costs)
    print("own times, iterations 0..2: mean", costs.mean(1), "max", costs.max(1))
else:
    costs = np.load(os.path.join(ROOT, "gpurun_out", "tile_costs.npy"))
    def snake_order(cost):
        order = np.zeros(1024, np.int32)
        for x in range(8):
            tiles = np.arange(x * 128, (x + 1) * 128)
            ranked = tiles[np.argsort(-cost[tiles], kind="stable")]
            for r, t in enumerate(ranked):
                rnd, pos = divmod(r, 32)
                cu = pos if rnd % 2 == 0 else 31 - pos
                order[x + 8 * (rnd * 32 + cu)] = t
        return order
    ctx = sv.Context(0, time_kernels=1)
    ctx.set_target(tgt, nrm)
    def measure(tag):
        acc = []
        for _ in range(6):
            ctx.set_source(src)
            ctx.run(6)
            f, _ = ctx.iteration_times()
            acc.append(np.asarray(f) * 1e3)
        X, _ = ctx.history()
        print(f"{tag}: launches 0..5 (median of 6 runs, us) {np.round(np.median(acc, 0), 2)}")
        return X
    for rep in range(2):
        ctx._check(ctx._lib.svicp_debug_set_tile_order(ctx._h, None, 0))
        X0 = measure("built-in order       ")
        for k in (0, 1):
            o = snake_order(costs[k])
            assert sorted(o.tolist()) == list(range(1024))
            ctx._check(ctx._lib.svicp_debug_set_tile_order(ctx._h, o.ctypes.data_as(C.POINTER(C.c_int32)), 1024))
            X1 = measure(f"by costs of launch {k} ")
            assert np.array_equal(X0, X1)
