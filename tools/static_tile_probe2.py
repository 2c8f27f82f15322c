"""Development probe: the sharpest proof "no point of this tile moves" that needs no coordinates.

A point p keeps its fp32 position under the increment (R, T) iff fl32(p + e) = p on every axis,
e = (R - I) p + T, i.e. iff |e_a| stays below half a unit in the last place of p_a (a quarter
below a power of two).  Nothing of an earlier launch's pre-rounding value enters: the update
starts from the ROUNDED coordinates, so the only per-tile quantities that can bound it are the
tile's coordinate box (for e, by interval arithmetic, signs and cancellation between the
rotation and the translation terms included) and the smallest |p_a| of the tile (for the unit in
the last place).  Prints per late iteration the share of tiles that really are static, the share
the box test proves, and what the looser test of tools/static_tile_probe.py proves.
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stereovision_amd as sv
from stereovision_amd import synth
from static_tile_probe import morton_tiles, rt_of   # noqa: E402  (runs that probe's module body: same pair)

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tgt, nrm, src, P = synth.make_pair(side, inlier_frac=0.7 if side > 1024 else None)
c = sv.Context(0)
c.set_target(tgt, nrm)
order = morton_tiles(src)
c.set_source(src)
c.run(12)
cur = c.get_source()
for k in range(12, 18):
    c.set_source(cur)
    c.run(2)
    nxt = c.get_source()
    Xh, _ = c.history()
    Rk, Tk = rt_of(Xh[0])
    moved = (nxt.view(np.uint32) != cur.view(np.uint32)).any(1)
    n = len(cur) // 1024 * 1024
    pt = cur[order[:n]].reshape(-1, 1024, 3).astype(np.float64)
    mv = moved[order[:n]].reshape(-1, 1024).any(1)
    lo, hi = pt.min(1), pt.max(1)
    E = Rk - np.eye(3)
    # interval of e_a over the tile's box
    a = lo[:, None, :] * E[None, :, :]; b = hi[:, None, :] * E[None, :, :]
    emin = np.minimum(a, b).sum(2) + Tk[None, :]
    emax = np.maximum(a, b).sum(2) + Tk[None, :]
    emag = np.maximum(np.abs(emin), np.abs(emax)) * (1 + 1e-9) + 1e-15       # fp64 rounding of the update itself
    mnn = np.abs(pt).min(1)
    with np.errstate(divide="ignore"):
        quarter_ulp = np.where(mnn > 0, 2.0 ** (np.floor(np.log2(np.maximum(mnn, 1e-300))) - 25), 0.0)
    box = (emag < 2 * quarter_ulp * 0.999).all(1)       # half an ulp (just under: ties to even)
    safe = (emag < quarter_ulp).all(1)                   # a quarter: also right below a power of two
    old = ((np.abs(pt).max(1) @ np.abs(E).T + np.abs(Tk)[None, :]) < quarter_ulp).all(1)
    print(f"update {k}: step |T| {np.abs(Tk).max():.1e} |R-I| {np.abs(E).max():.1e}; tiles {len(mv)}: "
          f"static in fact {1 - mv.mean():.3f}; proven by the box test at half an ulp {box.mean():.3f} "
          f"(of the static ones {box[~mv].mean():.3f}; wrongly {(box & mv).sum()}), at a quarter {safe.mean():.3f}, "
          f"by max|p| sums at a quarter {old.mean():.3f}", flush=True)
    cur = nxt
