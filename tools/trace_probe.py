"""Development probe for -DSVICP_TRACE builds: where one iteration's time goes.

Prints, per iteration, the gap to the previous launch, the span of the launch and the phase
boundaries of the block that ends last (all in microseconds from the launch's first stamp)."""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import stereovision_amd as sv
from stereovision_amd import synth, _lib

side = int(os.environ.get("SIDE", "1024"))
iters = int(os.environ.get("ITERS", "30"))
tgt, nrm, src, P = synth.make_pair(side)
ctx = sv.Context(0)
ctx.set_target(tgt, nrm)
ctx.set_source(src)
lib = ctypes.CDLL(_lib.LIB_PATH)
ctx.run(iters)
lib.svicp_debug_trace_clear()
ctx.run(iters)
NI, NT, NS = 64, 1024, 16
buf = np.zeros(NI * NT * NS, dtype=np.uint64)
lib.svicp_debug_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.svicp_debug_trace(buf.ctypes.data, buf.size) == 0
raw = buf.reshape(NI, NT, NS)
tr = raw[:, :, :8].astype(np.float64) * 0.01
NS = 8       # 100 MHz -> microseconds
prev_end = None
names = ["entry", "solved", "updated", "decided", "proved", "queued", "searched", "end"]
print("iter  gap  span | last block: " + " ".join(f"{n:>8s}" for n in names[1:]) + " | active tiles, median end of active, median 'decided' of all")
for k in range(iters):
    t = tr[k]
    t0 = t[:, 0][t[:, 0] > 0].min()
    last = np.where(t > 0, t, 0).max(axis=1)            # last stamp of each tile
    end = last.max()
    crit = int(last.argmax())
    active = t[:, 7] > 0
    row = t[crit]
    cells = " ".join(f"{(row[j] - t0):8.2f}" if row[j] > 0 else "       -" for j in range(1, NS))
    gap = (t0 - prev_end) if prev_end is not None else float("nan")
    med_act = np.median(t[active, 7] - t0) if active.any() else float("nan")
    dec = t[:, 2][t[:, 2] > 0]
    print(f"{k:3d} {gap:5.2f} {end - t0:6.2f} | {cells} | {int(active.sum()):4d} {med_act:6.2f} {np.median(dec - t0) if dec.size else float('nan'):6.2f}"
          f"  entry spread {np.percentile(t[:,0]-t0, 50):.2f}/{(t[:,0]-t0).max():.2f}")
    prev_end = end

# where the active tiles of a settled iteration run, and how their phases spread
for k in (17, 25):
    t = tr[k]
    t0 = t[:, 0].min()
    active = np.where(t[:, 7] > 0)[0]
    hw = raw[k, :, 8].astype(np.int64); xcc = raw[k, :, 9].astype(np.int64) & 15
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    place = xcc * 1000 + se * 100 + sh * 10 * 2 + cu      # one number per CU
    print(f"iteration {k}: active tiles {active.tolist()}")
    print("  XCD of active tiles:", np.bincount(xcc[active], minlength=8).tolist())
    per_cu = {}
    for a in active:
        per_cu.setdefault((int(xcc[a]), int(se[a]), int(sh[a]), int(cu[a])), []).append(int(a))
    sizes = np.array([len(v) for v in per_cu.values()])
    print("  CUs hosting active tiles:", len(per_cu), " active tiles per such CU: max", sizes.max(), "mean %.2f" % sizes.mean())
    dur = np.diff(t[active][:, :8], axis=1)
    for j, n in enumerate(names[1:]):
        d = dur[:, j]
        print(f"  phase -> {n:9s}: median {np.median(d):5.2f}  p90 {np.percentile(d, 90):5.2f}  max {d.max():5.2f}")
    ends = t[active, 7] - t0
    co = np.array([len(per_cu[(int(xcc[a]), int(se[a]), int(sh[a]), int(cu[a]))]) for a in active])
    for c in sorted(set(co.tolist())):
        print(f"  active tiles sharing their CU with {c - 1} other active tiles: {int((co == c).sum()):4d}, median end {np.median(ends[co == c]):.2f}, max end {ends[co == c].max():.2f}")

    sv_ = raw[k][active][:, 10:15].astype(np.float64) * 0.01 - t[active][:, 0:1]
    for j, n in enumerate(["sums loaded", "scaled", "factorised", "substituted", "R,T written"]):
        print(f"  solve: {n:12s} at median {np.median(sv_[:, j]):5.2f}  max {sv_[:, j].max():5.2f} after entry")

# how the dispatcher places blocks (iteration 0: every block does the same work)
hw = raw[0, :, 8].astype(np.int64); xcc = raw[0, :, 9].astype(np.int64) & 15
cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; wave = hw & 15
nb = 1024
print("block -> (xcc, se, sh, cu) for the blocks of XCD 0 in launch order")
line = []
for within in range(128):
    tile = 0 * 128 + within          # xcd_contiguous with nb = 1024: tile = xcd * 128 + within
    line.append(f"{within}:({xcc[tile]},{se[tile]},{sh[tile]},{cu[tile]})")
print(" ".join(line))
for it in (0, 17):
    hw = raw[it, :, 8].astype(np.int64); xcc = raw[it, :, 9].astype(np.int64) & 15
    cuid = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 15)
    print(f"iteration {it}: distinct CUs {len(set(cuid.tolist()))}, blocks per CU max {np.bincount(np.unique(cuid, return_inverse=True)[1]).max()}")

# which SIMD the first wave of each block sits on, for the blocks that share a CU
hw = raw[17, :, 8].astype(np.int64); xcc = raw[17, :, 9].astype(np.int64) & 15
cuid = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 15)
simd = (hw >> 4) & 3
from collections import Counter
pat = Counter()
for c in np.unique(cuid):
    pat[tuple(sorted(simd[cuid == c].tolist()))] += 1
print("SIMDs of the first waves of the blocks sharing a CU (pattern: count):", dict(pat))

# balance over the XCDs in the search-heavy iterations
for k in range(0, 6):
    t = tr[k]; t0 = t[:, 0].min()
    xcc = raw[k, :, 9].astype(np.int64) & 15
    last = np.where(t > 0, t, 0).max(axis=1) - t0
    srch = (t[:, 6] - t[:, 5])
    print(f"iteration {k}: per XCD max end " + " ".join(f"{last[xcc == x].max():6.2f}" for x in range(8)) +
          "   median end " + " ".join(f"{np.median(last[xcc == x]):6.2f}" for x in range(8)) +
          f"   search phase median {np.median(srch):.2f} p90 {np.percentile(srch, 90):.2f} max {srch.max():.2f}")

print("median over tiles, microseconds after the launch's first entry: entry | sums loaded, scaled, eliminated, substituted, R,T | solved updated decided proved queued searched end")
for k in range(1, 8):
    t = tr[k]; t0 = t[:, 0].min()
    act = t[:, 7] > 0
    sv_ = raw[k][:, 10:15].astype(np.float64) * 0.01 - t0
    ph = t[act][:, 1:8] - t0
    print(f"  {k}: {np.median(t[:, 0] - t0):5.2f} | " + " ".join(f"{np.median(sv_[:, j]):5.2f}" for j in range(5)) + " | " +
          " ".join(f"{np.median(ph[:, j]):5.2f}" for j in range(7)))

# the block that records (tile 0) against the other active tiles, and which tile ends a launch
for k in (17, 25, 29):
    t = tr[k]
    t0 = t[:, 0][t[:, 0] > 0].min()
    active = np.where(t[:, 7] > 0)[0]
    last = np.where(t > 0, t, 0).max(axis=1)
    others = [a for a in active if a != 0]
    print(f"iteration {k}: launch ends with tile {int(last.argmax())}; tile 0 (records): solved {t[0,1]-t0:.2f} end {t[0,7]-t0:.2f}; "
          f"other active tiles: solved median {np.median(t[others,1]-t0):.2f} end median {np.median(t[others,7]-t0):.2f} max {np.max(t[others,7]-t0):.2f}")
