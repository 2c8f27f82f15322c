"""Development probe: large sources with and without the per-tile proof "no point of this tile can
change" (svicp_debug_use_tile_boxes): same bits, time per late launch, tiles left unread."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth

iters = 30
for side in [int(a) for a in sys.argv[1:]] or [2048, 4096]:
    tgt, nrm, src, P = synth.make_pair(side, inlier_frac=0.7)
    c = sv.Context(0, time_kernels=1)
    c.set_target(tgt, nrm)
    out = {}
    for on in (0, 1, 0, 1):
        c._check(c._lib.svicp_debug_use_tile_boxes(c._h, on))
        c.set_source(src)
        c.run(iters)
        X, I = c.history()
        fused, _ = c.iteration_times()
        unread = c.tiles_left_unread()
        res = (X.copy(), I.copy(), c.get_source(), c.search_counts())
        if on in out:
            pass
        out.setdefault(on, res)
        print(f"side {side} boxes {on}: all {iters} launches {np.sum(fused) * 1e3:.0f} us, the first seven {np.round(np.asarray(fused[:7]) * 1e3, 0)}; "
              f"launches 20..29 {np.round(np.asarray(fused[20:]) * 1e3, 1)} us; "
              f"tiles {len(src) // 1024}, left unread per launch {unread[[2, 5, 8, 12, 20, 29]]}", flush=True)
    a, b = out[0], out[1]
    print("  same bits:", all(np.array_equal(x.view(np.uint8), y.view(np.uint8)) for x, y in zip(a, b)), flush=True)
    c.close()
