"""A seeded sweep over cloud shapes, scales and sizes the fixed cases do not hold: the HIP
search and one ICP iteration against the CPU oracle.  k-NN lists and correspondences (indices
and fp32 squared distances) bit-exact; the solution vector within 1e-8 scaled by the size of
the cloud's coordinates (the sums hold products of coordinates, so their rounding grows with
them).  SVICP_SWEEP_CASES in the environment widens the sweep (default 16 cases of each kind)."""
import os

import numpy as np
import pytest

import refmath

pytestmark = pytest.mark.gpu

CASES = int(os.environ.get("SVICP_SWEEP_CASES", "16"))
SIZES = [1, 2, 7, 63, 64, 65, 255, 1023, 1024, 1025, 2049, 5000, 12345, 40000]


def _unit(v):
    return v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)


def _shape(rng, kind, m, scale):
    """m points of one family, coordinates of the order of `scale`."""
    if kind == "sheet":                       # a bumpy height field, like a depth camera's cloud
        u = rng.random((m, 2)) * scale
        z = 0.05 * scale * (np.sin(u[:, 0] * 7 / scale) + np.cos(u[:, 1] * 5 / scale))
        return np.column_stack([u, z])
    if kind == "shell":
        return _unit(rng.normal(size=(m, 3))) * scale * 0.5 + scale * 0.5
    if kind == "volume":
        return rng.random((m, 3)) * scale
    if kind == "clusters":                    # dense lumps with empty space between them
        centres = rng.random((6, 3)) * scale
        return centres[rng.integers(0, 6, m)] + rng.normal(size=(m, 3)) * scale * 0.01
    if kind == "lattice":                     # exact ties in distance
        side = max(2, int(round(m ** (1 / 3))))
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)
        g = g[rng.permutation(len(g))[:m]] if len(g) >= m else g
        return g * (scale / side)
    if kind == "line":                        # all cells along one diagonal
        t = rng.random((m, 1)) * scale
        return t * np.float64([[1.0, 0.5, 0.25]])
    if kind == "repeats":                     # many coincident points
        base = rng.random((max(1, m // 5), 3)) * scale
        return base[rng.integers(0, len(base), m)]
    raise ValueError(kind)


KINDS = ["sheet", "shell", "volume", "clusters", "lattice", "line", "repeats"]


def _case(seed):
    rng = np.random.default_rng(1000 + seed)
    kind = KINDS[seed % len(KINDS)]
    m = int(rng.choice(SIZES[3:]))
    scale = float(rng.choice([0.5, 20.0, 300.0, 3000.0]))
    offset = rng.choice([0.0, 0.0, 1.0e4, -2.5e3]) * rng.random(3)
    tgt = (_shape(rng, kind, m, scale) + offset).astype(np.float32)
    return rng, kind, scale, np.ascontiguousarray(tgt)


@pytest.mark.parametrize("seed", range(CASES))
def test_knn_lists_bit_exact(ctx, po, seed):
    rng, kind, scale, tgt = _case(seed)
    m = len(tgt)
    ctx.set_target(tgt, np.zeros_like(tgt))
    tree = po.KDTree(tgt)
    nq = int(rng.choice(SIZES[:9]))
    pick = tgt[rng.integers(0, m, nq)].astype(np.float64)
    spread = rng.choice([0.0, 1e-3, 0.05, 2.0], size=(nq, 1)) * scale
    q = np.ascontiguousarray((pick + rng.normal(size=(nq, 3)) * spread).astype(np.float32))
    for k in sorted({1, int(rng.integers(2, 20)), 20, int(rng.integers(33, 60))}):
        gi, gd = ctx.knn(q, k)
        oi, od = tree.knn(q, k)
        kk = min(k, m)
        # equal distances may be listed in either order past the first of them only if the
        # oracle's own rule (lowest index first) is broken: compare everything
        assert np.array_equal(gd[:, :kk], od[:, :kk]), (kind, m, nq, k)
        assert np.array_equal(gi[:, :kk], oi[:, :kk]), (kind, m, nq, k)
        assert (gi[:, kk:] == -1).all() and np.isinf(gd[:, kk:]).all()


@pytest.mark.parametrize("seed", range(CASES))
def test_one_iteration_correspondences_and_solution(ctx, po, seed):
    """Iteration k on the oracle's state of iteration k (the inputs stay identical), three
    iterations: correspondences bit-exact, X and the inlier count against the oracle."""
    import stereovision_amd as sv
    rng = np.random.default_rng(5000 + seed)
    kind = ["sheet", "volume", "clusters", "sheet"][seed % 4]
    m = int(rng.choice(SIZES[6:]))
    n = int(rng.choice(SIZES[5:]))
    scale = float(rng.choice([40.0, 300.0, 1500.0]))
    offset = rng.choice([0.0, 0.0, 4.0e3]) * rng.random(3)
    tgt = np.ascontiguousarray((_shape(rng, kind, m, scale) + offset).astype(np.float32))
    nrm = np.ascontiguousarray(_unit(rng.normal(size=(m, 3))).astype(np.float32))
    # the source: target points displaced by less than the threshold's root, then moved rigidly
    pick = tgt[rng.integers(0, m, n)].astype(np.float64) + rng.normal(size=(n, 3)) * 0.3
    M = refmath.euler_to_RT(np.concatenate([rng.normal(size=3) * 2e-3, rng.normal(size=3) * 0.4]))
    centre = pick.mean(0)
    src = np.ascontiguousarray(refmath.apply_update((pick - centre).astype(np.float32), M) + centre.astype(np.float32))
    src = np.ascontiguousarray(src.astype(np.float32))
    ctx.set_target(tgt, nrm)
    tree = po.KDTree(tgt)
    big = float(np.abs(tgt).max()) + 1.0
    state = src
    for k in range(3):
        r = po.icp_point2plane(tree, nrm, state, 1)
        ctx.set_source(state)
        try:
            T, info = ctx.run(1)
        except sv.SvicpError as e:
            assert e.status == sv._lib.ERR_SINGULAR and (r["rc"] != 0 or r["info"][0, 0] < 12), (kind, m, n)
            return
        assert r["rc"] == 0, (kind, m, n)
        idx, d2 = ctx.correspondences()
        hit = r["d2"][0] <= 5.0
        assert np.array_equal(idx[hit], r["idx"][0][hit]) and np.array_equal(d2[hit], r["d2"][0][hit]), (kind, m, n, k)
        assert (idx[~hit] == -1).all() and np.isinf(d2[~hit]).all()
        X, I = ctx.history()
        assert I[0, 0] == r["info"][0, 0] == hit.sum()
        if hit.sum() >= 100:                                   # fewer rows: the system is ill-conditioned
            assert np.abs(X[0] - r["X"][0]).max() < 1e-8 * big, (kind, m, n, k, hit.sum())
        state = refmath.apply_update(state, refmath.euler_to_RT(r["X"][0]))
