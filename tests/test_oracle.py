"""CPU tests: the oracle against the golden vectors and against independent
methods (brute force, scipy cKDTree, numpy.linalg).  The reference holds no
expected outputs for this path, so none can be checked here: parity unpinned."""
import numpy as np
import pytest

import refmath


def test_parse_matches_loadtxt(po, tmp_path):
    text = b"1.5 2.25 -3\n 4  5 6 \n7 8 9 10\n"
    got = po.parse_txt(text)
    # tokens: 1.5 2.25 -3 4 5 6 7 8 9 10 -> 10 tokens, 3 rows (CTxtReader.cpp:51)
    assert got.shape == (3, 3)
    assert np.array_equal(got, np.float32([[1.5, 2.25, -3], [4, 5, 6], [7, 8, 9]]))
    from stereovision_amd import CTxtReader
    p = tmp_path / "c.txt"
    p.write_bytes(text)
    assert np.array_equal(CTxtReader().read(str(p)), got)
    assert CTxtReader().read(str(tmp_path / "missing.txt")).shape == (0, 3)
    # the binary form round-trips the same cloud (and drops a trailing partial triple)
    b = tmp_path / "c.bin"
    CTxtReader.writeBinary(str(b), got)
    assert np.array_equal(CTxtReader().readBinary(str(b)), got)
    with open(b, "ab") as fh:
        fh.write(np.float32([1.0]).tobytes())
    assert np.array_equal(CTxtReader().readBinary(str(b)), got)
    assert CTxtReader().readBinary(str(tmp_path / "missing.bin")).shape == (0, 3)


def test_kdtree_equals_brute_random(po):
    rng = np.random.default_rng(1)
    ref = rng.normal(size=(3000, 3)).astype(np.float32) * 10
    q = rng.normal(size=(500, 3)).astype(np.float32) * 12
    tree = po.KDTree(ref)
    for k in (1, 5, 20, 128):
        i1, d1 = tree.knn(q, k)
        i2, d2 = po.knn_brute(ref, q, k)
        assert np.array_equal(i1, i2) and np.array_equal(d1, d2)


def test_kdtree_ties_and_duplicates(po, lattice):
    tgt, q = lattice["target"], lattice["query"]
    tree = po.KDTree(tgt)
    i1, d1 = tree.knn(q, 8)
    assert np.array_equal(i1, lattice["idx"]) and np.array_equal(d1, lattice["d2"])
    # a cell-centre query has 8 equidistant corners: ascending index order
    assert np.all(np.diff(i1[0]) > 0) and np.all(d1[0] == d1[0][0])
    dup = np.concatenate([tgt, tgt])
    i2, _ = po.KDTree(dup).knn(tgt[:10], 2)
    assert np.array_equal(i2[:, 0], np.arange(10)) and np.array_equal(i2[:, 1], np.arange(10) + len(tgt))


def test_knn_more_than_points(po):
    ref = np.float32([[0, 0, 0], [1, 0, 0], [0, 2, 0]])
    i, d = po.KDTree(ref).knn(np.float32([[0.1, 0, 0]]), 5)
    assert list(i[0]) == [0, 1, 2, -1, -1] and np.isinf(d[0, 3:]).all()


def test_kdtree_vs_scipy_on_bundled_pair(po, pair):
    from scipy.spatial import cKDTree
    ref, src, gold = pair
    i1, d1 = po.KDTree(ref).knn(src, 1)
    dd, ii = cKDTree(ref.astype(np.float64)).query(src.astype(np.float64), 1)
    assert np.array_equal(ii, i1[:, 0])
    assert np.allclose(dd ** 2, d1[:, 0], rtol=1e-5, atol=1e-5)
    assert np.array_equal(i1[:, 0], gold["nn_idx"]) and np.array_equal(d1[:, 0], gold["nn_d2"])


def test_eig3_vs_numpy(po):
    rng = np.random.default_rng(2)
    for _ in range(200):
        A = rng.normal(size=(3, 3))
        W = A @ A.T + np.outer(rng.normal(size=3), rng.normal(size=3)) * 0
        c = rng.normal(size=3) * 100
        W = W + np.outer(c, c)
        v, ev = po.eig3_smallest(W)
        w, V = np.linalg.eigh(W)
        assert abs(ev - w[0]) <= 1e-9 * abs(w[2])
        assert abs(abs(v @ V[:, 0]) - 1.0) < 1e-9


def test_normals_golden_and_restatement(po, pair):
    ref, _, gold = pair
    tree = po.KDTree(ref)
    normals, knn = po.calc_normals(tree, want_knn=True)
    assert np.array_equal(normals, gold["normals"])
    assert np.array_equal(knn.astype(np.int64).sum(axis=1), gold["knn20_rowsum"])
    assert np.array_equal(knn[:64], gold["knn20_head"])
    assert np.array_equal(knn[:, 0], np.arange(len(ref)))       # self is its own nearest
    # numpy restatement of CPoint2PlaneICP.cpp:132-151 on a few points
    for p in (0, 17, 5000, 11793):
        pts = ref[knn[p]]
        W = np.zeros((3, 3))
        for row in pts:
            W += ((np.outer(row, row).astype(np.float32)) / np.float32(20)).astype(np.float32).astype(np.float64)
        w, V = np.linalg.eigh(W)
        n = V[:, 0]
        assert abs(abs(n @ normals[p].astype(np.float64)) - 1.0) < 1e-6


def test_icp_golden(po, pair):
    ref, src, gold = pair
    tree = po.KDTree(ref)
    r = po.icp_point2plane(tree, gold["normals"], src, 5)
    assert r["rc"] == 0
    assert np.array_equal(r["idx"], gold["idx5"])
    assert np.allclose(r["X"], gold["X5"], rtol=0, atol=1e-13)
    assert np.allclose(r["trans"], gold["trans5"], rtol=0, atol=1e-13)
    assert np.array_equal(r["src"], gold["src5"])
    # the bundled pair is an exact rigid motion: Rx(1 deg) Ry(0.5 deg), t = (1, 0, 0)
    assert np.allclose(r["trans"][:3, 3], [1, 0, 0], atol=1e-4)
    r30 = po.icp_point2plane(tree, gold["normals"], src, 30, history=False)
    assert np.allclose(r30["trans"], gold["trans30"], rtol=0, atol=1e-12)


def test_icp_expression_order_is_immaterial(po, pair):
    ref, src, gold = pair
    tree = po.KDTree(ref)
    a = po.icp_point2plane(tree, gold["normals"], src, 3, faithful=True)
    b = po.icp_point2plane(tree, gold["normals"], src, 3, faithful=False)
    assert np.allclose(a["X"], b["X"], rtol=0, atol=1e-11)


def test_single_iteration_vs_numpy(po, synth32):
    tgt, nrm, src = synth32["target"], synth32["normals"], synth32["source"]
    tree = po.KDTree(tgt)
    r = po.icp_point2plane(tree, nrm, src, 2)
    idx, d2 = tree.knn(src, 1)
    assert np.array_equal(idx[:, 0], synth32["idx0"])
    S = refmath.sums32(src, tgt, nrm, idx[:, 0], d2[:, 0])
    X = refmath.solve_from_sums(S)
    assert np.allclose(X, r["X"][0], rtol=0, atol=1e-10)
    assert r["info"][0, 0] == S[27]
    assert np.isclose(r["info"][0, 1], S[28], rtol=1e-12)
    M = refmath.euler_to_RT(X)
    src1 = refmath.apply_update(src, refmath.euler_to_RT(r["X"][0]))
    # state after one update is what iteration 1 searched from
    idx1, _ = tree.knn(src1, 1)
    assert np.array_equal(idx1[:, 0], r["idx"][1])
    assert np.allclose(M, refmath.euler_to_RT(r["X"][0]), atol=1e-9)


def test_synthetic_recovers_pose(po, synth32):
    tgt, nrm, src, P = synth32["target"], synth32["normals"], synth32["source"], synth32["pose"]
    r = po.icp_point2plane(po.KDTree(tgt), nrm, src, 10)
    assert np.allclose(r["trans"], synth32["trans10"], rtol=0, atol=1e-12)
    Pinv = np.linalg.inv(P)
    assert refmath.rot_angle(r["trans"][:3, :3], Pinv[:3, :3]) < 2e-3
    # in-plane sliding on a small smooth patch is weakly constrained, so the
    # translation is checked through the point-to-plane residual it leaves
    assert r["info"][-1, 1] < 0.01 * r["info"][0, 1]
    assert r["info"][-1, 0] >= 0.95 * len(src)


def test_empty_source_is_singular(po, synth32):
    tgt, nrm = synth32["target"], synth32["normals"]
    r = po.icp_point2plane(po.KDTree(tgt), nrm, np.zeros((0, 3), np.float32), 1)
    assert r["rc"] != 0 and np.isnan(r["X"]).all()


# ---- point-to-point ICP (CPoint2PointICP.cpp) -----------------------------------
def test_svd3_and_kabsch_against_numpy(po):
    rng = np.random.default_rng(0)
    for trial in range(200):
        W = rng.normal(size=(3, 3)) * rng.uniform(0.1, 100)
        if trial % 3 == 0:
            W[:, 2] *= 1e-6
        U, S, V = po.svd3(W)
        assert np.allclose(U @ np.diag(S) @ V.T, W, atol=1e-12 * np.abs(W).max())
        assert np.allclose(V.T @ V, np.eye(3), atol=1e-12) and np.all(np.diff(S) <= 0)
        u, s, vt = np.linalg.svd(W)
        assert np.allclose(S, s, rtol=1e-10)
        D = np.diag([1, 1, np.sign(np.linalg.det(vt.T @ u.T))])
        assert np.allclose(po.kabsch_rotation(W), vt.T @ D @ u.T, atol=1e-8)


def _p2p_single_numpy(ref, src, idx, d2, thr=15.0):
    """One CPoint2PointICP::singleRun (:83-159) in numpy, centroid sums in fp64."""
    F = np.float32
    d2 = d2.astype(F)
    sl = (d2 / F(thr)).astype(F)
    ss = (sl * sl).astype(F)
    w = ((1.0 - ss.astype(np.float64)) ** 2).astype(F)
    sel = d2 < F(thr)
    w, mv, rf = w[sel], src[sel].astype(F), ref[idx[sel]].astype(F)
    sw = w.astype(np.float64).sum()
    mu_m = ((mv * w[:, None]).astype(F).astype(np.float64).sum(0) / sw).astype(F)
    mu_r = ((rf * w[:, None]).astype(F).astype(np.float64).sum(0) / sw).astype(F)
    mc, rc = (mv - mu_m).astype(F), (rf - mu_r).astype(F)
    count = F(sel.sum())
    W = np.zeros((3, 3))
    for j in range(3):
        for i in range(3):
            prod = (mc[:, j] * rc[:, i]).astype(F)
            W[j, i] = ((w * prod).astype(F) / count).astype(F).astype(np.float64).sum()
    u, s, vt = np.linalg.svd(W)
    D = np.diag([1, 1, np.sign(np.linalg.det(vt.T @ u.T))])
    R = vt.T @ D @ u.T
    T = mu_r.astype(np.float64) - R @ mu_m.astype(np.float64)
    return R, T, W, int(sel.sum())


def test_point2point_first_iteration_against_numpy(po, pair):
    ref, src, _ = pair
    tree = po.KDTree(ref)
    r = po.icp_point2point(tree, src, 1, faithful=False)
    R, T, W, count = _p2p_single_numpy(ref, src, r["idx"][0], r["d2"][0])
    assert r["info"][0, 0] == count
    assert np.allclose(r["W"][0], W, rtol=1e-12, atol=1e-12 * np.abs(W).max())
    assert np.allclose(r["RT"][0, :9].reshape(3, 3), R, atol=1e-10)
    assert np.allclose(r["RT"][0, 9:], T, atol=1e-8)


def test_point2point_registers_the_bundled_pair(po, pair):
    """17.txt is 15.txt moved rigidly: five iterations bring the weighted squared distance to
    (almost) zero.  The fp32 running sums of the reference (faithful) and fp64 sums differ in
    the fifth digit of the translation."""
    ref, src, _ = pair
    tree = po.KDTree(ref)
    a = po.icp_point2point(tree, src, 5, faithful=True)
    b = po.icp_point2point(tree, src, 5, faithful=False)
    assert a["rc"] == 0 and b["rc"] == 0
    assert a["info"][-1, 0] == len(src) and a["info"][-1, 2] < 1e-2 and b["info"][-1, 2] < 1e-2
    assert np.abs(a["trans"] - b["trans"]).max() < 1e-4
    assert refmath.rot_angle(a["trans"][:3, :3], b["trans"][:3, :3]) < 1e-6
    d = np.abs(a["src"] - ref[a["idx"][-1]]).max()
    assert d < 1e-2


@pytest.mark.parametrize("side", [192, 512])
def test_synthetic_pair_with_inlier_fraction(side):
    """make_pair(inlier_frac=0.7): at least 70 % of the source starts within the weight
    function's threshold (CPoint2PlaneICP.cpp:68) of a target point, at any extent; the
    default pose of the same pair leaves far fewer once the cloud is large."""
    from scipy.spatial import cKDTree
    from stereovision_amd import synth
    tgt, _, src, P = synth.make_pair(side, inlier_frac=0.7)
    d, _ = cKDTree(tgt).query(src)
    frac = float((d * d <= 5.0).mean())
    assert 0.70 <= frac <= 0.85, frac
    assert np.allclose(P[:3, :3] @ P[:3, :3].T, np.eye(3), atol=1e-12)
    if side == 512:
        _, _, src0, _ = synth.make_pair(side)
        d0, _ = cKDTree(tgt).query(src0)
        assert (d0 * d0 <= 5.0).mean() < 0.40


def test_unpinned_third_party_bits_cannot_move_the_transform(po, pair):
    """The reference's FLANN and Eigen are not available, so the last bits of its estimated
    normals (SelfAdjointEigenSolver, CPoint2PlaneICP.cpp:148) and of each solve
    (MatrixXd::inverse, :109) are unknown.  This bounds what they could change: every normal
    component moved by one fp32 ulp up or down (random signs, re-normalisation left out, as a
    differently rounded eigenvector would be), and every solve's X displaced by 1e-12 per
    component, move the 5- and the 30-iteration transform of the bundled pair by less than
    1e-7 rad and 1e-7 x extent: a tenth of the tolerance this path is held to."""
    ref, src, gold = pair
    tree = po.KDTree(ref)
    normals = gold["normals"]
    extent = float(np.ptp(ref, axis=0).max())
    rng = np.random.default_rng(20251111)
    worst_rot, worst_tr, flips = 0.0, 0.0, 0
    for iters in (5, 30):
        base = po.icp_point2plane(tree, normals, src, iters)
        for trial in range(3):
            up = rng.integers(0, 2, normals.shape).astype(bool)
            moved = np.where(up, np.nextafter(normals, np.float32(2.0)), np.nextafter(normals, np.float32(-2.0)))
            assert np.abs(moved.view(np.int32) - normals.view(np.int32)).max() == 1
            po.set_solve_perturbation(1e-12, seed=trial + 1)
            try:
                r = po.icp_point2plane(tree, moved.astype(np.float32), src, iters)
            finally:
                po.set_solve_perturbation(0.0)
            worst_rot = max(worst_rot, refmath.rot_angle(r["trans"][:3, :3], base["trans"][:3, :3]))
            worst_tr = max(worst_tr, float(np.abs(r["trans"][:3, 3] - base["trans"][:3, 3]).max()))
            flips += int((r["idx"] != base["idx"]).sum())
    print(f"worst rotation change {worst_rot:.2e} rad, translation {worst_tr:.2e} "
          f"(extent {extent:.0f}); {flips} correspondences differ over all runs")
    assert worst_rot < 1e-7 and worst_tr < 1e-7 * extent
