"""GPU parity tests: the HIP path through the C ABI against the CPU oracle and
the golden vectors.  Integer results (correspondence indices, k-NN lists) and
fp32 distances must be bit-exact; the transform must agree within 1e-6 in
rotation (radians) and 1e-6 * extent in translation, the tolerance the
specification of this path states.  In practice agreement is ~1e-10."""
import numpy as np
import pytest

import refmath

pytestmark = pytest.mark.gpu

ROT_TOL = 1e-6
X_TOL = 1e-8


def _check_transform(T, T_ref, extent):
    assert refmath.rot_angle(T[:3, :3], T_ref[:3, :3]) < ROT_TOL
    assert np.abs(T[:3, 3] - T_ref[:3, 3]).max() < 1e-6 * extent
    assert np.array_equal(T[3], [0, 0, 0, 1])


def _extent(cloud):
    return float((cloud.max(0) - cloud.min(0)).max())


# ---------------------------------------------------------------- bundled pair
def test_normals_match_oracle(ctx, pair):
    ref, _, gold = pair
    ctx.set_target(ref)
    n = ctx.get_normals()
    sign = np.sign((n * gold["normals"]).sum(1))[:, None]
    assert np.abs(n * sign - gold["normals"]).max() <= 1.2e-7      # one fp32 ulp of a unit vector
    assert (np.abs((n * gold["normals"]).sum(1)) > 0.999999).all()


def test_centred_normals_option_is_the_covariance_estimate(ctx, pair, po):
    """centred_normals = 1 has no counterpart in the reference (CPoint2PlaneICP.cpp:140-146 sums
    moments about the origin): checked against numpy's eigenvectors of the 20-neighbourhood
    covariance, and on a synthetic surface against the analytic normals, where the reference's
    estimate is visibly the worse one."""
    from stereovision_amd import synth
    ref, _, gold = pair
    ctx.set_options(centred_normals=1)
    try:
        ctx.set_target(ref)
        n = ctx.get_normals()
        idx, _ = po.KDTree(ref).knn(ref, 20)
        nb = ref[idx].astype(np.float64)                        # m x 20 x 3
        d = nb - nb.mean(1, keepdims=True)
        w, v = np.linalg.eigh(np.einsum("mki,mkj->mij", d, d))
        want = v[:, :, 0]
        clear = (w[:, 1] - w[:, 0]) > 1e-3 * w[:, 2]            # smallest eigenvalue well separated
        cosines = np.abs((n * want).sum(1))
        assert clear.mean() > 0.95 and (cosines[clear] > 1 - 1e-6).all()
        tgt, nrm, _, _ = synth.make_pair(256)
        ctx.set_target(tgt)
        centred = np.abs((ctx.get_normals() * nrm).sum(1))
        ctx.set_options(centred_normals=0)
        ctx.set_target(tgt)
        as_reference = np.abs((ctx.get_normals() * nrm).sum(1))
        assert np.median(centred) > 0.995 and np.median(centred) > np.median(as_reference)
    finally:
        ctx.set_options(centred_normals=0)


def test_non_finite_coordinates(ctx):
    """The reference validates nothing (NaNs would wander through FLANN and Eigen).  Here an
    infinite coordinate is refused, and a NaN point is inert: a NaN source point finds no
    neighbour, a NaN target point is nobody's neighbour, the other points register as before."""
    import stereovision_amd as sv
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(48)
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    T0, info0 = ctx.run(8)
    bad = tgt.copy()
    bad[5, 1] = np.inf
    with pytest.raises(sv.SvicpError):
        ctx.set_target(bad, nrm)
    t2, s2 = tgt.copy(), src.copy()
    t2[[7, 100, 2000], :] = np.nan
    s2[[3, 50, 1500], 0] = np.nan
    ctx.set_target(t2, nrm)
    ctx.set_source(s2)
    T1, info1 = ctx.run(8)
    assert info1.status == 0 and abs(info1.inliers - info0.inliers) <= 12
    assert np.abs(T1 - T0).max() < 0.05
    idx, d2 = ctx.correspondences()
    assert (idx[[3, 50, 1500]] == -1).all() and np.isinf(d2[[3, 50, 1500]]).all()
    assert not np.isin(idx, [7, 100, 2000]).any()


def test_contexts_on_several_host_threads():
    """One object per thread at a time, objects independent (as the reference's classes): four
    threads making, using and destroying their own contexts at once get the serial results."""
    import threading
    import stereovision_amd as sv
    from stereovision_amd import synth
    pairs = [synth.make_pair(64 + 16 * i, pair=i) for i in range(4)]

    def register(i):
        tgt, nrm, src, _ = pairs[i]
        with sv.Context(0) as c:
            c.set_target(tgt, nrm)
            c.set_source(src)
            T = c.run(10)[0]
            c.knn(src[:100], 5)
            return T

    serial = [register(i) for i in range(4)]
    out = [[] for _ in range(4)]

    def work(i):
        for _ in range(6):
            out[i].append(register(i))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(len(out[i]) == 6 and all(np.array_equal(T, serial[i]) for T in out[i]) for i in range(4))


def test_knn_bit_exact_on_bundled_pair(ctx, pair, po):
    ref, src, gold = pair
    ctx.set_target(ref, gold["normals"])
    idx, d2 = ctx.knn(src, 1)
    assert np.array_equal(idx[:, 0], gold["nn_idx"]) and np.array_equal(d2[:, 0], gold["nn_d2"])
    tree = po.KDTree(ref)
    for k in (2, 20, 33):
        gi, gd = ctx.knn(ref, k)
        oi, od = tree.knn(ref, k)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)
    assert np.array_equal(ctx.knn(ref, 20)[0][:64], gold["knn20_head"])


@pytest.mark.parametrize("graph", [0, 1])
def test_icp_5_iterations_as_test_cpp(ctx, pair, graph):
    ref, src, gold = pair
    ctx.set_options(use_graph=graph)
    ctx.set_target(ref)                       # normals estimated on the device
    ctx.set_source(src)
    T, info = ctx.run(5)                      # test.cpp:23
    X, I = ctx.history()
    _check_transform(T, gold["trans5"], _extent(ref))
    assert np.abs(X - gold["X5"]).max() < X_TOL
    assert info.iterations == 5 and info.status == 0
    assert info.inliers == gold["info5"][-1, 0]
    assert np.allclose(I[:, 0], gold["info5"][:, 0])
    assert np.allclose(I[:, 1], gold["info5"][:, 1], rtol=1e-6, atol=1e-9)
    # the source is left transformed count-1 times (CPoint2PlaneICP.cpp:42-50)
    out = ctx.get_source()
    assert np.abs(out - gold["src5"]).max() <= 4e-5          # a few fp32 ulps at |x| ~ 250
    assert (out == gold["src5"]).mean() > 0.99
    T30, _ = ctx.run(30)
    _check_transform(T30, gold["trans30"], _extent(ref))


def test_correspondences_bit_exact_every_iteration(ctx, pair, po):
    """Iteration k searches from the cloud after k updates.  Feeding the oracle's
    state of each iteration keeps the inputs identical, so indices must match bit
    for bit wherever a neighbour lies within the threshold."""
    ref, src, gold = pair
    ctx.set_target(ref, gold["normals"])
    tree = po.KDTree(ref)
    state = src.copy()
    for k in range(5):
        if k > 0:
            state = refmath.apply_update(state, refmath.euler_to_RT(gold["X5"][k - 1]))
        ctx.set_source(state)
        T, info = ctx.run(1)
        X, _ = ctx.history()
        assert np.abs(X[0] - gold["X5"][k]).max() < X_TOL
        idx, d2 = ctx.correspondences()
        oi, od = tree.knn(state, 1)
        hit = od[:, 0] <= 5.0
        assert np.array_equal(idx[hit], oi[hit, 0]) and np.array_equal(d2[hit], od[hit, 0])
        assert (idx[~hit] == -1).all() and np.isinf(d2[~hit]).all()
        if k < 4:
            assert np.array_equal(oi[:, 0], gold["idx5"][k])
        assert np.array_equal(ctx.get_source(), state)       # one solve: no update applied


def test_compat_class_updates_caller_buffer(pair, capsys):
    import stereovision_amd as sv
    ref, src, gold = pair
    icp = sv.CPoint2PlaneICP()
    icp.setReferenceData(ref)
    move = src.copy()
    icp.setMoveData(move)
    icp.setMaxIterateCount(5)
    icp.run()
    assert "Current Transformation:" in capsys.readouterr().out
    _check_transform(icp.getTransform(), gold["trans5"], _extent(ref))
    assert np.abs(move - gold["src5"]).max() <= 4e-5 and not np.array_equal(move, src)
    # a second run continues from the updated cloud, as the reference would
    icp.setMaxIterateCount(1)
    icp.run()
    assert np.abs(icp.getTransform()[:3, 3]).max() < 1e-3


def test_knn_class(pair, po):
    import stereovision_amd as sv
    ref, src, gold = pair
    knn = sv.CKNNSearch()
    knn.setTreeData(ref)
    knn.setQueryData(src[:500])
    knn.setKCount(200)                       # clamps to 128 (CKNNSearch.cpp:60-65)
    idx, d2 = knn.getKNeighbour()
    assert idx.shape == (500, 128)
    oi, od = po.KDTree(ref).knn(src[:500], 128)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)


def test_cpp_driver_runs_like_test_cpp(pair, tmp_path):
    """examples/register_pair.cpp makes the calls of the reference's driver (test.cpp:16-24)
    through the drop-in classes; it must print the golden transform."""
    import os
    import subprocess
    ref, src, gold = pair
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "test_icp")
    assert os.path.exists(exe), "build it with make -C stereovision_amd/csrc example"
    np.savetxt(tmp_path / "17.txt", ref, fmt="%.8f")
    np.savetxt(tmp_path / "15.txt", src, fmt="%.8f")
    out = subprocess.run([exe, str(tmp_path / "17.txt"), str(tmp_path / "15.txt")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    at = lines.index("Current Transformation:")
    T = np.array([[float(v) for v in lines[at + 1 + r].split()] for r in range(4)])
    assert np.abs(T - gold["trans5"]).max() < 1e-5          # printed with 6 significant digits
    assert "Euler Angle X Y Z:" in out.stdout and "searchint spend" in out.stdout
    e = lines.index("Euler Angle X Y Z: ")
    ang = [float(lines[e + 1 + k]) for k in range(3)]
    # the bundled pair differs by Rx(1 deg) Ry(0.5 deg); eulerAngles(0,1,2) reports R = Rx Ry Rz
    assert abs(abs(ang[0]) - 1.0) < 1e-2 or abs(abs(ang[0]) - 179.0) < 1e-2


def test_coherent_search_matches_full_search(ctx, pair):
    """Skipping the search for points whose old correspondence is provably still
    the nearest neighbour must not change anything."""
    from stereovision_amd import synth
    ref, src, gold = pair
    cases = [(ref, gold["normals"], src)]
    tgt, nrm, s, _ = synth.make_pair(100)
    cases.append((tgt, nrm, s))
    for tgt, nrm, s in cases:
        res = []
        for coh in (0, 1):
            ctx.set_options(coherent=coh)
            ctx.set_target(tgt, nrm)
            ctx.set_source(s)
            T, info = ctx.run(12)
            X, I = ctx.history()
            res.append((T, X, I, ctx.get_source(), ctx.correspondences()))
        (T0, X0, I0, s0, c0), (T1, X1, I1, s1, c1) = res
        assert np.abs(X0 - X1).max() < 1e-11 and np.abs(T0 - T1).max() < 1e-11
        assert np.array_equal(I0[:, 0], I1[:, 0])              # same inlier counts every iteration
        assert np.abs(s0 - s1).max() <= 4e-5
        same = (s0 == s1).all(axis=1)
        assert same.mean() > 0.999
        assert np.array_equal(c0[0][same], c1[0][same]) and np.array_equal(c0[1][same], c1[1][same])
    ctx.set_options(coherent=1)


@pytest.mark.parametrize("side", [2048, 2100])
def test_large_source_leaves_settled_tiles_unread(ctx, side):
    """4096 tiles (the smallest source that counts as large) and 4307 (the last tile partly filled,
    the tiles not a multiple of the blocks): the launches from the fifth on run
    one round of blocks that work through several tiles each, deal out the tiles that had work in
    the launch before and leave unread those whose box records prove that no point of them can
    change.  With the proofs switched off every tile is read; the results must be the same
    bits, and agree with a run that searches every point in every iteration."""
    from stereovision_amd import synth
    tgt, nrm, src, P = synth.make_pair(side, inlier_frac=0.7)
    assert (len(src) + 1023) // 1024 >= 4096
    ctx.set_target(tgt, nrm)
    res = []
    for boxes, coherent in ((1, 1), (0, 1), (1, 0)):
        ctx._check(ctx._lib.svicp_debug_use_tile_boxes(ctx._h, boxes))
        ctx.set_options(coherent=coherent)
        ctx.set_source(src)
        T, info = ctx.run(24)
        X, I = ctx.history()
        res.append((T, X, I, ctx.get_source(), ctx.search_counts(), ctx.tiles_left_unread()))
    ctx._check(ctx._lib.svicp_debug_use_tile_boxes(ctx._h, 1))
    ctx.set_options(coherent=1)
    on, off, full = res
    tiles = (len(src) + 1023) // 1024
    assert on[5][:4].max() == 0 and on[5][-1] > 0.8 * tiles and off[5].max() == 0 and full[5].max() == 0
    for a, b in zip(on[:5], off[:5]):
        assert np.array_equal(a.view(np.uint8), b.view(np.uint8))
    assert np.abs(on[0] @ P - np.eye(4)).max() < 5e-4
    assert np.abs(on[1] - full[1]).max() < 1e-10 and np.abs(on[0] - full[0]).max() < 1e-10
    assert np.array_equal(on[2][:, 0], full[2][:, 0])
    assert ((on[3] == full[3]).all(axis=1)).mean() > 0.999


# ---------------------------------------------------------------- synthetic
@pytest.mark.parametrize("side", [32, 100, 256])
def test_synthetic_pair_matches_oracle(ctx, po, side):
    from stereovision_amd import synth
    tgt, nrm, src, P = synth.make_pair(side)
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    T, info = ctx.run(6)
    X, I = ctx.history()
    r = po.icp_point2plane(po.KDTree(tgt), nrm, src, 6)
    _check_transform(T, r["trans"], _extent(tgt))
    assert np.abs(X - r["X"]).max() < X_TOL
    assert np.array_equal(I[:, 0], r["info"][:, 0])
    idx, d2 = ctx.correspondences()
    # states can differ by an fp32 ulp after several fp64-summed solves; indices still agree
    hit = r["d2"][-1] <= 4.99
    assert (idx[hit] == r["idx"][-1][hit]).mean() > 0.9999


def test_unsorted_source_gives_same_sums(ctx, po, synth32):
    tgt, nrm, src = synth32["target"], synth32["normals"], synth32["source"]
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    T1, _ = ctx.run(3)
    ctx.set_options(sort_source=0)
    ctx.set_source(src)
    T2, _ = ctx.run(3)
    assert np.abs(T1 - T2).max() < 1e-10
    perm = np.random.default_rng(0).permutation(len(src))
    ctx.set_source(np.ascontiguousarray(src[perm]))
    T3, _ = ctx.run(3)
    assert np.abs(T1 - T3).max() < 1e-10
    out = ctx.get_source()
    ctx.set_options(sort_source=1)
    ctx.set_source(src)
    ctx.run(3)
    assert np.abs(out - ctx.get_source()[perm]).max() <= 4e-5


# ---------------------------------------------------------------- edge cases
def test_ties_resolve_to_lowest_index(ctx, lattice):
    ctx.set_target(lattice["target"], np.zeros_like(lattice["target"]))
    idx, d2 = ctx.knn(lattice["query"], 8)
    assert np.array_equal(idx, lattice["idx"]) and np.array_equal(d2, lattice["d2"])
    i1, _ = ctx.knn(lattice["query"], 1)
    assert np.array_equal(i1[:, 0], lattice["idx"][:, 0])


def test_duplicates_and_k_larger_than_target(ctx):
    tgt = np.float32([[0, 0, 0], [1, 0, 0], [0, 0, 0], [0, 2, 0]])
    ctx.set_target(tgt, np.zeros_like(tgt))
    idx, d2 = ctx.knn(np.float32([[0.1, 0, 0]]), 6)
    assert list(idx[0]) == [0, 2, 1, 3, -1, -1]
    assert np.isinf(d2[0, 4:]).all() and d2[0, 0] == d2[0, 1]


def test_knn_dense_sparse_and_coincident_targets(ctx, po):
    """The wave-per-query search: a cell box with far more candidates than one round
    takes (several rounds, each keeping the best), boxes with fewer than k points (wider
    rings), more coincident points than a wave can keep apart (passed on to the general
    search), and queries off the target."""
    rng = np.random.default_rng(11)
    dense = (rng.random((40000, 3)) * [30, 30, 4]).astype(np.float32)          # ~3000 per box
    sparse = (rng.random((300, 3)) * 400).astype(np.float32)
    heap = np.repeat(np.float32([[5, 5, 5], [6, 5, 5]]), [100, 3], axis=0)      # 100 coincident
    for tgt in (dense, sparse, np.concatenate([dense[:2000], heap])):
        ctx.set_target(tgt, np.zeros_like(tgt))
        tree = po.KDTree(tgt)
        q = np.concatenate([tgt[:300], (tgt[:100] + rng.normal(size=(100, 3)) * 3).astype(np.float32),
                            np.float32([[5, 5, 5.01], [-50, 10, 2]])])
        for k in (1, 2, 7, 20, 32):
            gi, gd = ctx.knn(q, k)
            oi, od = tree.knn(q, k)
            assert np.array_equal(gi, oi) and np.array_equal(gd, od), (len(tgt), k)


def test_far_queries_use_the_exhaustive_pass(ctx, po, synth32):
    tgt = synth32["target"]
    ctx.set_target(tgt, synth32["normals"])
    rng = np.random.default_rng(5)
    q = (tgt[rng.integers(0, len(tgt), 64)] + rng.normal(size=(64, 3)) * 300).astype(np.float32)
    q[0] = [1e6, -1e6, 1e6]
    tree = po.KDTree(tgt)
    for k in (1, 5):
        gi, gd = ctx.knn(q, k)
        oi, od = tree.knn(q, k)
        assert np.array_equal(gi, oi) and np.array_equal(gd, od)


def test_no_overlap_is_reported_singular(ctx, synth32):
    import stereovision_amd as sv
    ctx.set_target(synth32["target"], synth32["normals"])
    ctx.set_source(synth32["source"] + np.float32(1000.0))
    with pytest.raises(sv.SvicpError) as e:
        ctx.run(2)
    assert e.value.status == sv._lib.ERR_SINGULAR
    idx, d2 = ctx.correspondences()
    assert (idx == -1).all()


def test_empty_and_tiny_inputs(ctx, synth32):
    import stereovision_amd as sv
    tgt, nrm = synth32["target"], synth32["normals"]
    ctx.set_target(tgt, nrm)
    ctx.set_source(np.zeros((0, 3), np.float32))
    with pytest.raises(sv.SvicpError):
        ctx.run(1)
    assert ctx.get_source().shape == (0, 3)
    idx, d2 = ctx.knn(np.zeros((0, 3), np.float32), 3)
    assert idx.shape == (0, 3)
    ctx.set_source(tgt[:1].copy())
    with pytest.raises(sv.SvicpError):        # one row cannot fix six unknowns
        ctx.run(1)
    ctx.set_target(np.zeros((0, 3), np.float32))
    idx, d2 = ctx.knn(tgt[:4], 2)
    assert (idx == -1).all() and np.isinf(d2).all()
    with pytest.raises(sv.SvicpError):
        ctx.run(0)


def test_call_order_is_checked():
    import stereovision_amd as sv
    c = sv.Context(0)
    with pytest.raises(sv.SvicpError) as e:
        c.run(1)
    assert e.value.status == sv._lib.ERR_ARG
    with pytest.raises(sv.SvicpError):
        c.knn(np.zeros((1, 3), np.float32), 1)
    c.close()


# ---------------------------------------------------------------- full size
def test_one_million_points_properties(ctx):
    """The benchmark's configuration: 1024^2 points, 30 iterations.  The oracle
    needs ~1 s per iteration here, so this checks size-independent properties:
    the recovered motion undoes the generating pose, returned distances are the
    distances to the returned indices, no sampled point is closer, and a
    shuffled source gives the same answer."""
    from stereovision_amd import synth
    tgt, nrm, src, P = synth.make_pair(1024)
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    T, info = ctx.run(30)
    Pinv = np.linalg.inv(P)
    assert refmath.rot_angle(T[:3, :3], Pinv[:3, :3]) < 2e-5
    moved = src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    truth = src.astype(np.float64) @ Pinv[:3, :3].T + Pinv[:3, 3]
    assert np.sqrt(((moved - truth) ** 2).sum(1)).max() < 0.05
    assert info.inliers > 0.98 * len(src) and info.step_norm < 1e-4

    idx, d2 = ctx.correspondences()
    cur = ctx.get_source()
    hit = idx >= 0
    assert hit.mean() > 0.98 and (d2[hit] <= 5.0).all()
    diff = cur[hit] - tgt[idx[hit]]
    re = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]).astype(np.float32)
    re = (re + diff[:, 2] * diff[:, 2]).astype(np.float32)
    assert np.array_equal(re, d2[hit])
    rng = np.random.default_rng(3)
    rows = rng.integers(0, len(cur), 2000)
    cand = rng.integers(0, len(tgt), 4000)
    dd = ((cur[rows, None, :].astype(np.float64) - tgt[None, cand, :]) ** 2).sum(-1).min(1)
    ok = np.where(idx[rows] >= 0, d2[rows] <= dd * (1 + 1e-6), dd > 5.0 * (1 - 1e-6))
    assert ok.all()

    perm = rng.permutation(len(src))
    ctx.set_source(np.ascontiguousarray(src[perm]))
    T2, _ = ctx.run(30)
    assert refmath.rot_angle(T[:3, :3], T2[:3, :3]) < 1e-9
    assert np.abs(T[:3, 3] - T2[:3, 3]).max() < 1e-6


def test_one_million_points_twenty_neighbours(ctx, po):
    """The normal estimation's search at the benchmark's size: the 20 nearest of every target
    point in the target itself.  Properties of every row (the point itself first at distance
    zero, distances ascending, ties by index, the distances those of the named points), rows
    drawn at random against the oracle's tree, and the normals estimated from the rows against
    the surface's own."""
    from stereovision_amd import synth
    tgt, nrm, _, _ = synth.make_pair(1024)
    ctx.set_target(tgt)                                   # estimates the normals: 20-NN + eigenvectors
    idx, d2 = ctx.knn(tgt, 20)
    assert idx.shape == (len(tgt), 20) and (idx >= 0).all() and (idx < len(tgt)).all()
    assert np.array_equal(idx[:, 0], np.arange(len(tgt))) and (d2[:, 0] == 0).all()
    assert (np.diff(d2, axis=1) >= 0).all()
    tie = np.diff(d2, axis=1) == 0
    assert (np.diff(idx, axis=1)[tie] > 0).all()
    rows = np.random.default_rng(9).integers(0, len(tgt), 3000)
    diff = tgt[rows, None, :] - tgt[idx[rows]]
    re = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]).astype(np.float32)
    re = (re + diff[..., 2] * diff[..., 2]).astype(np.float32)
    assert np.array_equal(re, d2[rows])
    oi, od = po.KDTree(tgt).knn(tgt[rows], 20)
    assert np.array_equal(idx[rows], oi) and np.array_equal(d2[rows], od)
    # uncentred moments (the reference's rule) tilt the estimate on a cloud far from the
    # origin; it still has to follow the surface
    est = ctx.get_normals()
    assert np.median(np.abs((est * nrm).sum(1))) > 0.95


@pytest.mark.parametrize("side,extra", [(1024, 0), (195, 11), (40, 3)])
def test_block_order_takes_every_tile_once(ctx, side, extra):
    """Once few tiles still have work, the tiles that had some take the first blocks of the
    next launch.  Whatever the order, every tile has to be taken exactly once: the sums, the
    inlier counts and the increments must equal those of a run without the shortcuts (where
    the order is fixed and every point is searched), on a full-size pair (1024 tiles), a
    ragged one (38 tiles, the last one partial) and a small one, and again on the same
    context after a source of another size."""
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(side)
    src = src[: len(src) - extra] if extra else src
    ctx.set_target(tgt, nrm)
    runs = []
    for coh, s in ((0, src), (1, src), (1, src[: len(src) // 3]), (1, src)):
        ctx.set_options(coherent=coh)
        ctx.set_source(np.ascontiguousarray(s))
        T, info = ctx.run(30)
        X, I = ctx.history()
        runs.append((T, X, I))
    ctx.set_options(coherent=1)
    (T0, X0, I0), (T1, X1, I1), _, (T3, X3, I3) = runs
    for T, X, I in ((T1, X1, I1), (T3, X3, I3)):
        assert np.array_equal(I0[:, 0], I[:, 0])                   # inliers, every iteration
        assert np.abs(X0 - X).max() < 1e-10 and np.abs(T0 - T).max() < 1e-10
        assert np.abs(I0[:, 3] - I[:, 3]).max() <= 1e-9 * I0[:, 3].max()   # sums of weights


def test_exact_and_capped_search_agree_within_threshold(ctx, po):
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(128)
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    capped_i, capped_d = ctx.correspondences()
    exact_i, exact_d = ctx.knn(src, 1)
    oi, od = po.KDTree(tgt).knn(src, 1)
    assert np.array_equal(exact_i, oi) and np.array_equal(exact_d, od)
    hit = od[:, 0] <= 5.0
    assert np.array_equal(capped_i[hit], oi[hit, 0]) and (capped_i[~hit] == -1).all()
    assert ctx.time_nn(exact=True, reps=2) > 0 and ctx.time_nn(exact=False, reps=2) > 0


def test_one_neighbour_search_equals_first_of_two(ctx, po):
    """svicp_knn with k = 1 runs the thread-per-query exact search (what svicp_time_nn times as
    the exact search), k = 2 the wave-per-query one: two kernels, the same first neighbour, at
    the start pose (most queries have to look beyond the capped probe's cube, whose runs cover
    quarter cells, not whole cells) and against the oracle."""
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(256)
    ctx.set_target(tgt, nrm)
    rng = np.random.default_rng(3)
    far = (src[:4000] + rng.normal(size=(4000, 3)) * [3, 3, 12]).astype(np.float32)
    q = np.concatenate([src, far])
    i1, d1 = ctx.knn(q, 1)
    i2, d2 = ctx.knn(q, 2)
    assert np.array_equal(i1[:, 0], i2[:, 0]) and np.array_equal(d1[:, 0], d2[:, 0])
    oi, od = po.KDTree(tgt).knn(q[::7], 1)
    assert np.array_equal(i1[::7], oi) and np.array_equal(d1[::7], od)
    assert (d1[:, 0] > 5.2).mean() > 0.3          # beyond the probe's cube for many
    # a query with a NaN coordinate: no neighbour with k = 1; with k > 1 the distances are NaN
    bad = np.float32([[np.nan, 0, 0], [1, 2, np.nan]])
    bi, bd = ctx.knn(np.concatenate([bad, src[:3]]), 1)
    assert (bi[:2] == -1).all() and np.isinf(bd[:2]).all() and (bi[2:, 0] >= 0).all()
    bi, bd = ctx.knn(np.concatenate([bad, src[:3]]), 2)
    assert np.isnan(bd[:2]).all() and (bi[2:, 0] >= 0).all()


def test_probe_records_that_do_not_fit_the_packed_form(ctx, po):
    """Probe records are packed in 16 bytes (run lengths in a byte, the distance between the
    starts of rows next to each other in 16 bits); the records that do not fit are kept in
    full.  A pile of near-coincident points (runs of more than 254) and two long parallel
    lines of points one cell apart (rows more than 65535 points apart in the sorted target)
    make such records; the capped search, the loop and the exact search must answer as the
    oracle does, with and without them."""
    rng = np.random.default_rng(23)
    pile = (np.float32([10, 10, 10]) + rng.random((700, 3)) * 0.05).astype(np.float32)
    xs = np.linspace(-200.0, 3000.0, 70000)
    line_a = np.stack([xs, np.full_like(xs, 40.0), np.full_like(xs, 5.0)], 1)
    # four lines 2.5 apart in y: every row of cells (about 4.5 wide) they cross holds one or two
    lines = [line_a + [0.013 * k, 2.5 * k, 0.0] for k in range(4)]
    line_b = lines[3]
    cloud = (rng.random((3000, 3)) * [60, 60, 20]).astype(np.float32)
    tgt = np.concatenate([cloud, pile] + lines).astype(np.float32)
    nrm = np.zeros_like(tgt); nrm[:, 2] = 1
    ctx.set_target(tgt, nrm)
    ranked, rec_bytes, in_full = ctx.grid_info()
    assert in_full > 0 and rec_bytes == 16 * 256 * ranked + 32 * in_full
    q = np.concatenate([pile[:200] + np.float32(0.3), cloud[:500] + np.float32(0.2),
                        line_a[::400] + [0.2, 1.9, 0.3], line_b[::400] + [-0.2, -1.7, 0.1],
                        np.float32([[10.02, 10.02, 10.02], [1400, 42.3, 5], [500, 500, 500]])]).astype(np.float32)
    ctx.set_source(q)
    ci, cd = ctx.correspondences()
    oi, od = po.KDTree(tgt).knn(q, 1)
    hit = od[:, 0] <= 5.0
    assert hit.sum() > 800 and (~hit).sum() >= 1
    assert np.array_equal(ci[hit], oi[hit, 0]) and np.array_equal(cd[hit], od[hit, 0])
    assert (ci[~hit] == -1).all()
    ei, ed = ctx.knn(q, 1)
    assert np.array_equal(ei, oi) and np.array_equal(ed, od)
    # a second target in the same context without such records: none are counted
    ctx.set_target(cloud, np.zeros_like(cloud))
    assert ctx.grid_info()[2] == 0
    ctx.set_source(cloud[:100] + np.float32(0.1))
    ci, _ = ctx.correspondences()
    oi, _ = po.KDTree(cloud).knn(cloud[:100] + np.float32(0.1), 1)
    assert np.array_equal(ci, oi[:, 0])


# ---- batches of independent pairs ---------------------------------------------
def _batch_pairs():
    """Pairs of different sizes (ragged last tiles, fewer points than one tile) and poses."""
    from stereovision_amd import synth
    out = []
    for i, (side, cut) in enumerate([(64, 0), (48, 7), (96, 1000), (20, 0), (64, 3), (33, 0)]):
        tgt, nrm, src, _ = synth.make_pair(side, pair=i, pose=synth.small_pose(i))
        out.append((tgt, nrm, np.ascontiguousarray(src[: len(src) - cut])))
    return out


@pytest.mark.parametrize("coherent", [1, 0])
def test_batch_equals_one_by_one(po, coherent):
    import stereovision_amd as sv
    pairs = _batch_pairs()
    singles, ctxs = [], []
    for tgt, nrm, src in pairs:
        c = sv.Context(0, coherent=coherent)
        c.set_target(tgt, nrm)
        c.set_source(src)
        T, info = c.run(8)
        singles.append((T, info.inliers, c.history()[0], c.get_source()))
        c.set_source(src)              # back to the start for the batch
        ctxs.append(c)
    batch = sv.Batch(ctxs)
    for _ in range(2):                 # the second run replays the captured graph
        Ts, infos = batch.run(8)
        for i, c in enumerate(ctxs):
            T, inl, X, moved = singles[i]
            assert np.abs(Ts[i] - T).max() < 1e-10
            assert infos[i].inliers == inl and infos[i].iterations == 8
            assert np.abs(c.history()[0] - X).max() < 1e-10
            assert np.abs(c.get_source() - moved).max() <= 4e-5
    # and against the oracle for one of them
    tgt, nrm, src = pairs[2]
    r = po.icp_point2plane(po.KDTree(tgt), nrm, src, 8)
    assert np.abs(Ts[2] - r["trans"]).max() < 1e-6 * max(1.0, np.abs(r["trans"]).max())
    batch.close()


def test_batch_argument_errors():
    import stereovision_amd as sv
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(16)
    a, b = sv.Context(0), sv.Context(0)
    a.set_target(tgt, nrm)
    a.set_source(src)
    with pytest.raises(sv.SvicpError):
        sv.Batch([a, b])               # b holds nothing
    with pytest.raises(sv.SvicpError):
        sv.Batch([a, a])
    with pytest.raises(sv.SvicpError):
        sv.Batch([a]).run(0)


def test_batch_many_small_pairs_throughput_property():
    """64 pairs of 128^2 points: every pair converges to its own pose, and the batch does
    the work of 64 single runs."""
    import stereovision_amd as sv
    from stereovision_amd import synth
    ctxs, poses = [], []
    for i in range(64):
        tgt, nrm, src, pose = synth.make_pair(128, pair=i, pose=synth.small_pose(i))
        c = sv.Context(0)
        c.set_target(tgt, nrm)
        c.set_source(src)
        ctxs.append(c)
        poses.append(pose)
    batch = sv.Batch(ctxs)
    Ts, infos = batch.run(30)
    for i in range(64):
        assert infos[i].status == 0 and infos[i].inliers > 0.9 * 128 * 128
        assert infos[i].step_norm < 1e-3
    # the recovered motion undoes the pose the source was moved by
    for i in (0, 17, 63):
        M = Ts[i] @ poses[i]
        assert refmath.rot_angle(M[:3, :3], np.eye(3)) < 2e-3 and np.abs(M[:3, 3]).max() < 0.5
    ctxs[5].set_source(synth.make_pair(128, pair=5, pose=synth.small_pose(5))[2])
    T0, _ = ctxs[5].run(30)
    assert np.abs(T0 - Ts[5]).max() < 1e-10
    print(f"batch of 64 x 16384 points, 30 iterations: {batch.last_ms():.3f} ms")


@pytest.mark.parametrize("side,iters,frac", [(1024, 4, None), (4096, 2, None), (1024, 4, 0.7), (4096, 2, 0.7)])
def test_full_size_iterations_against_oracle(ctx, po, side, iters, frac):
    """The benchmark's 1M-point pair and the 16.8M-point pair of the sharded configuration, at
    the fixed start pose and at the pose that leaves 70 % of the source within the threshold,
    first iterations, against the oracle on all host threads: inlier counts equal, solve
    results within 1e-9 (rotation) / 1e-9 x extent (translation)."""
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(side, inlier_frac=frac)
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    T, info = ctx.run(iters)
    X, I = ctx.history()
    searched = ctx.search_counts()
    r = po.icp_point2plane(po.KDTree(tgt), nrm, src, iters, threads=po.max_threads(), history=False)
    extent = float(np.ptp(tgt, axis=0).max())
    assert np.array_equal(I[:, 0], r["info"][:, 0])                 # inliers per iteration
    assert np.abs(X[:, :3] - r["X"][:, :3]).max() < 1e-9
    assert np.abs(X[:, 3:] - r["X"][:, 3:]).max() < 1e-9 * extent
    assert np.abs(T - r["trans"]).max() < 1e-9 * extent
    assert searched[0] == len(src) and np.all(searched[1:] <= len(src))
    assert np.abs(ctx.get_source() - r["src"]).max() <= 2e-4        # a few ulp at |x| ~ 1e4


@pytest.mark.parametrize("frac", [None, 0.7])
def test_one_million_points_thirty_iterations_against_oracle(ctx, po, frac):
    """The headline configuration from end to end (CPoint2PlaneICP.cpp:40-57: 30 solves, 29 updates
    of the source), both start poses, against the oracle on all host threads.  Iterations 5 to 29
    are the settled regime -- tiles skipped, correspondences proven instead of searched, blocks
    reordered, near-tie points looked ahead for -- so every shortcut is held against a run that
    takes none.  Inliers equal in every iteration, every solve within 1e-9 (angles) and
    1e-9 x extent (translation), the accumulated transform within the 1e-6 / 1e-6 x extent this
    path is held to (measured: far inside), the moved source within an fp32 ulp."""
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(1024, inlier_frac=frac)
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    T, info = ctx.run(30)
    X, I = ctx.history()
    searched = ctx.search_counts()
    r = po.icp_point2plane(po.KDTree(tgt), nrm, src, 30, threads=po.max_threads(), history=False)
    extent = float(np.ptp(tgt, axis=0).max())
    assert np.array_equal(I[:, 0], r["info"][:, 0])
    assert np.abs(X[:, :3] - r["X"][:, :3]).max() < 1e-9
    assert np.abs(X[:, 3:] - r["X"][:, 3:]).max() < 1e-9 * extent
    Rg, Ro = T[:3, :3], r["trans"][:3, :3]
    assert np.abs(Rg - Ro).max() < 1e-6 and np.abs(T[:3, 3] - r["trans"][:3, 3]).max() < 1e-6 * extent
    assert np.abs(T - r["trans"]).max() < 1e-9 * extent            # (what is measured: 1e-12 .. 1e-10)
    # the settled regime was really exercised
    assert searched[0] == len(src) and searched[8:].max() < 0.001 * len(src)
    assert np.abs(ctx.get_source() - r["src"]).max() <= 2.5e-4      # one ulp at |x| ~ 2600


def test_search_counts_fall_as_the_registration_settles(ctx):
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(256)
    ctx.set_target(tgt, nrm)
    ctx.set_source(src)
    ctx.run(12)
    s = ctx.search_counts()
    assert s[0] == len(src) and s[-1] < 0.01 * len(src)
    ctx.set_options(coherent=0)
    ctx.set_source(src)
    ctx.run(4)
    assert np.all(ctx.search_counts() == len(src))
    ctx.set_options(coherent=1)


@pytest.mark.parametrize("thr,k", [(2.0, 20), (12.5, 8), (5.0, 33)])
def test_options_other_than_the_reference_constants(po, pair, thr, k):
    """dist_threshold and normals_k are options of the library (the reference hard-codes 5 and
    20, CPoint2PlaneICP.cpp:68, :125); the oracle is told the same numbers."""
    import stereovision_amd as sv
    ref, src, _ = pair
    po.set_constants(thr, k)
    try:
        tree = po.KDTree(ref)
        normals = po.calc_normals(tree)
        want = po.icp_point2plane(tree, normals, src, 6)
    finally:
        po.set_constants()
    c = sv.Context(0, dist_threshold=thr, normals_k=k)
    c.set_target(ref)
    c.set_source(src)
    T, info = c.run(6)
    X, I = c.history()
    got_n = c.get_normals()
    flip = np.sign((got_n * normals).sum(1))
    assert np.abs(got_n * flip[:, None] - normals).max() < 2e-6
    assert np.array_equal(I[:, 0], want["info"][:, 0])
    assert np.abs(X - want["X"]).max() < 1e-7
    _check_transform(T, want["trans"], _extent(ref))


def test_history_rows_come_from_the_library(ctx, synth32):
    import stereovision_amd as sv
    ctx.set_target(synth32["target"], synth32["normals"])
    ctx.set_source(synth32["source"])
    ctx.run(5)
    with pytest.raises(sv.SvicpError):
        ctx.run(0)                                   # refused: the last successful run stays 5
    assert ctx.last_iterations() == 5
    X, I = ctx.history()
    assert X.shape == (5, 6) and I.shape == (5, 4) and np.isfinite(X).all()
    assert ctx.search_counts().shape == (5,)
    # arrays smaller than the run are refused instead of overrun
    import ctypes as C
    small = np.zeros((3, 6))
    rc = ctx._lib.svicp_get_history(ctx._h, 3, small.ctypes.data_as(C.POINTER(C.c_double)), None)
    assert rc == sv._lib.ERR_ARG and not small.any()
    rc = ctx._lib.svicp_search_counts(ctx._h, 4, small.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc == sv._lib.ERR_ARG and not small.any()
    ctx.run_point2point(4)
    RT, I = ctx.history_rt()
    assert RT.shape == (4, 12)


def test_default_stream_runs_without_capture(ctx, synth32):
    """The legacy default stream cannot be captured into a graph: a context told to launch
    on it runs the loop eagerly and returns the same transform."""
    ctx.set_target(synth32["target"], synth32["normals"])
    ctx.set_source(synth32["source"])
    T_own, _ = ctx.run(6)
    ctx.set_stream(0)
    T_null, info = ctx.run(6)
    Tp_null, _ = ctx.run_point2point(3)
    ctx.use_own_stream()
    T_again, _ = ctx.run(6)
    Tp_own, _ = ctx.run_point2point(3)
    assert info.status == 0
    assert np.abs(T_null - T_own).max() < 1e-10 and np.abs(T_again - T_own).max() < 1e-10
    assert np.abs(Tp_null - Tp_own).max() < 1e-9


def test_arena_batch_equals_one_by_one(po):
    """svicp_batch_create_from_arrays on ragged pairs: every pair's result equals its own
    svicp_run and the oracle; borrowed contexts give the moved source and the history."""
    import stereovision_amd as sv
    from stereovision_amd import synth
    sides = [48, 96, 33, 64, 80]
    T_one, tg, nr, sr = [], [], [], []
    for i, side in enumerate(sides):
        t, n, s, _ = synth.make_pair(side, pair=10 + i, pose=synth.small_pose(i))
        s = s[: len(s) - 7 * i]                        # ragged sources
        tg.append(t); nr.append(n); sr.append(s)
        with sv.Context(0) as c:
            c.set_target(t, n)
            c.set_source(s)
            T, info = c.run(9)
            T_one.append((T, info.inliers, c.get_source(), c.history()[0]))
    batch = sv.ArenaBatch(tg, nr, sr)
    assert batch.info()[0] == len(sides) and batch.info()[2] > 0
    for rep in range(2):                               # the second run replays the graph
        Ts, infos = batch.run(9)
        for i in range(len(sides)):
            assert np.abs(Ts[i] - T_one[i][0]).max() < 1e-10
            assert infos[i].inliers == T_one[i][1] and infos[i].status == 0
    for i in (0, 3):
        c = batch.pair(i)
        assert np.array_equal(c.get_source(), T_one[i][2])
        assert np.abs(c.history()[0] - T_one[i][3]).max() < 1e-10
        with pytest.raises(sv.SvicpError):
            c.knn(sr[i][:4], 1)                        # the exact-search tables are not kept
        c.close()                                      # a borrowed context: nothing is destroyed
    r = po.icp_point2plane(po.KDTree(tg[1]), nr[1], sr[1], 9)
    assert np.abs(Ts[1] - r["trans"]).max() < 1e-8
    batch.close()
    # normals estimated per pair when none are given
    b2 = sv.ArenaBatch(tg[:2], None, sr[:2])
    T2, _ = b2.run(5)
    with sv.Context(0) as c:
        c.set_target(tg[1]); c.set_source(sr[1])
        assert np.abs(T2[1] - c.run(5)[0]).max() < 1e-10
    b2.close()
    with pytest.raises(sv.SvicpError):
        sv.ArenaBatch(tg[:1], nr[:1], sr[:1], dist_threshold=-1.0)


def test_config4_slice_512_pairs_of_65536_points(po):
    """The per-GPU slice of the batched configuration at its stated size: 512 independent pairs
    of 65536 points, 30 iterations, one arena.  Every pair recovers its generating pose (a
    property that holds at any size); three sampled pairs against the oracle."""
    import stereovision_amd as sv
    from stereovision_amd import synth
    count, side, iters = 512, 256, 30
    tg, nr, sr, poses = [], [], [], []
    for i in range(count):
        t, n, s, P = synth.make_pair(side, pair=i, pose=synth.small_pose(i))
        tg.append(t); nr.append(n); sr.append(s); poses.append(P)
    batch = sv.ArenaBatch(tg, nr, sr)
    Ts, infos = batch.run(iters)
    ms_first = batch.last_ms()
    Ts, infos = batch.run(iters)
    n, setup_ms, dev_bytes = batch.info()
    print(f"512 x 65536 points, 30 iterations: set-up {setup_ms:.0f} ms, run {batch.last_ms():.2f} ms "
          f"(first {ms_first:.2f}), arena {dev_bytes / 2**30:.2f} GiB")
    for i in range(count):
        assert infos[i].status == 0 and infos[i].inliers > 0.9 * side * side
        E = Ts[i] @ poses[i]                           # the transform undoes the generating pose
        assert refmath.rot_angle(E[:3, :3], np.eye(3)) < 2e-4
        assert np.abs(E[:3, 3]).max() < 0.05
    for i in (0, 255, 511):
        r = po.icp_point2plane(po.KDTree(tg[i]), nr[i], sr[i], iters, threads=po.max_threads(), history=False)
        assert np.abs(Ts[i] - r["trans"]).max() < 1e-8
    batch.close()


def test_two_runs_give_the_same_bits():
    """The sums are kept in 64-bit fixed point, so the order in which the tiles' atomic
    additions arrive cannot change them: two runs on the 1 M-point pair return bit-identical
    solution vectors, transforms and moved sources, and so does a second context."""
    import stereovision_amd as sv
    from stereovision_amd import synth
    tgt, nrm, src, _ = synth.make_pair(1024)
    outs = []
    for _ in range(2):
        with sv.Context(0) as c:
            c.set_target(tgt, nrm)
            c.set_source(src)
            for rep in range(2):
                T, info = c.run(30)
                outs.append((T.copy(), c.history()[0].copy(), c.get_source().copy(), info.inliers))
    for T, X, moved, inl in outs[1:]:
        assert np.array_equal(T, outs[0][0])
        assert np.array_equal(X, outs[0][1])
        assert np.array_equal(moved, outs[0][2])
        assert inl == outs[0][3]
