"""GPU parity of the calibration reprojection path (include/svcal.h) against the CPU oracle.

Everything is fp64 on both sides.  The residuals go through the same operations in the same
order, but the device's sin/cos/sqrt/divide may differ from libm's in the last bit, and a
central difference with eps = 1e-6 amplifies a one-ulp difference of a ~500 pixel value
(1.1e-13) to about 6e-8 in a Jacobian entry.  Tolerances: residuals 1e-9 pixel absolute;
normal-equation blocks 1e-6 relative to the largest entry of the block (a Jacobian-entry
error of 6e-8 against entries of order 1 to 1e3); |e|^2 1e-10 relative.
"""
import os

import numpy as np
import pytest

from stereovision_amd import synth
from tests.refmath import homography_start, stereo_start

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
BLOCKS = ("LL", "RR", "RS", "SS", "LE", "EE", "gL", "gR", "gE", "gS")


def _compare(got, want, names=BLOCKS):
    assert np.abs(got["residual"] - want["residual"]).max() < 1e-9
    assert got["sq_residual"] == pytest.approx(want["sq_residual"], rel=1e-10)
    for n in names:
        scale = np.abs(want[n]).max()
        assert scale > 0
        err = np.abs(got[n] - want[n]).max() / scale
        assert err < 1e-6, f"block {n}: {err}"


@pytest.mark.parametrize("views,k", [(1, 5), (3, 5), (3, 2), (64, 5)])
def test_stereo_blocks_match_oracle(po, views, k):
    from stereovision_amd.calib import ReprojectionEvaluator
    w, l, r, _, start = synth.make_stereo_views(views=views)
    if k == 2:
        start = start[np.r_[0:6, 9:15, 18:len(start)]]
    ev = ReprojectionEvaluator(w, l, r)
    got = ev.evaluate(start, k)
    want = po.stereo_eval_sparse(start, k, w, l, r)
    _compare(got, want)
    ev.close()


def test_stereo_many_corners_and_ragged_chunks(po):
    """More corners than threads in a block (130 = 128 + 2): the per-view loop takes a second,
    nearly empty chunk."""
    from stereovision_amd.calib import ReprojectionEvaluator
    w, l, r, _, start = synth.make_stereo_views(views=5, board=(13, 10))
    ev = ReprojectionEvaluator(w, l, r)
    _compare(ev.evaluate(start), po.stereo_eval_sparse(start, 5, w, l, r))


def test_stereo_full_size(po):
    """2000 views of an 11 x 8 board: 704,000 residual rows, 12,024 parameters."""
    from stereovision_amd.calib import ReprojectionEvaluator
    w, l, r, _, start = synth.make_stereo_views(views=2000)
    assert len(start) == 12024 and 4 * l.shape[0] * l.shape[1] == 704000
    ev = ReprojectionEvaluator(w, l, r)
    got = ev.evaluate(start)
    want = po.stereo_eval_sparse(start, 5, w, l, r)
    _compare(got, want)
    # size-independent property: the sums over views of the per-view blocks of a two-copy
    # problem are twice those of the one-copy problem
    ev2 = ReprojectionEvaluator(w, np.concatenate([l, l]), np.concatenate([r, r]))
    nE = 6 * 2000
    p2 = np.r_[start[:18], start[18:18 + nE], start[18:18 + nE], start[-6:]]
    got2 = ev2.evaluate(p2, want_residual=False)
    for n in ("LL", "RR", "RS", "SS", "gL", "gR", "gS"):
        assert np.allclose(got2[n], 2.0 * got[n], rtol=1e-12, atol=1e-12 * np.abs(got[n]).max())
    assert np.array_equal(got2["EE"][:2000], got2["EE"][2000:])
    assert np.array_equal(got2["EE"][:2000], got["EE"])


def test_reference_corners_stereo_and_refinement(po):
    """The reference's own stereo corner tables (Test/testdata.h) through the reference's
    damped Gauss-Newton loop, evaluated on the device; the oracle runs the same steps."""
    from stereovision_amd.calib import ReprojectionEvaluator, assemble_dense, lm_refine
    c = np.load(os.path.join(GOLD, "testdata_corners.npz"))
    w, l, r = c["world_stereo"], c["left"], c["right"]
    x0 = stereo_start(w, l, r, (640, 480), 535.0)
    ev = ReprojectionEvaluator(w, l, r)
    _compare(ev.evaluate(x0), po.stereo_eval_sparse(x0, 5, w, l, r))
    x, err = lm_refine(ev, x0, 5, max_iter=40)
    rms = err / np.sqrt(4 * 4 * 54)
    assert rms < 0.6
    # the oracle agrees on the error at the refined parameters
    e = po.stereo_residual(x, 5, w, l, r)
    assert np.sqrt(e @ e) == pytest.approx(err, rel=1e-9)
    # and one oracle step from the same point matches one device step
    b_dev = ev.evaluate(x0)
    b_cpu = po.stereo_eval_sparse(x0, 5, w, l, r)
    A1, G1 = assemble_dense(b_dev, 5, 4)
    A2, G2 = assemble_dense(b_cpu, 5, 4)
    s1 = np.linalg.solve(A1 + 1e-3 * np.diag(np.diag(A1)), G1)
    s2 = np.linalg.solve(A2 + 1e-3 * np.diag(np.diag(A2)), G2)
    assert np.allclose(s1, s2, rtol=1e-4, atol=1e-6 * np.abs(s2).max())


def test_mono_matches_oracle(po):
    from stereovision_amd.calib import ReprojectionEvaluator, assemble_dense
    c = np.load(os.path.join(GOLD, "testdata_corners.npz"))
    w, img = c["world_mono"], c["mono"]
    x0 = homography_start(w, img, (1920, 1080))
    ev = ReprojectionEvaluator(w, img)
    got = ev.evaluate(x0)
    res, J = po.mono_residual_jacobian_dense(x0, w, img)
    assert np.abs(got["residual"] - res).max() < 1e-9
    A, G = assemble_dense(got, 5, 4, stereo=False)
    JtJ, Jte = J.T @ J, J.T @ res
    assert np.abs(A - JtJ).max() / np.abs(JtJ).max() < 1e-6
    assert np.abs(G - Jte).max() / np.abs(Jte).max() < 1e-6
    assert got["sq_residual"] == pytest.approx(res @ res, rel=1e-10)


def test_argument_errors():
    from stereovision_amd import SvicpError
    from stereovision_amd.calib import ReprojectionEvaluator
    w, l, r, _, start = synth.make_stereo_views(views=2)
    ev = ReprojectionEvaluator(w, l, r)
    with pytest.raises(SvicpError):
        ev.eval()                           # no parameters yet
    with pytest.raises(ValueError):
        ev.set_params(start[:-1])
    with pytest.raises(SvicpError):
        ev.set_params(np.zeros(ev.n_params(3)), 3)   # k must be 2 or 5
    mono = ReprojectionEvaluator(w, l)
    with pytest.raises(SvicpError):
        mono.set_params(np.zeros(mono.n_params(2)), 2)


def test_two_thousand_views_refine_to_the_noise_floor(po):
    """End to end at the full size: the reference's loop with the block-sparse solve brings a
    perturbed start (rms of a few pixels) down to the pixel noise the data was generated with
    (0.2 pixels); the reference's dense inverse of a 12,024 x 12,024 matrix per step is what
    this replaces.  The oracle confirms the final error."""
    from stereovision_amd.calib import ReprojectionEvaluator, lm_refine
    w, l, r, truth, start = synth.make_stereo_views(views=2000, perturb=1e-2)
    ev = ReprojectionEvaluator(w, l, r)
    rows = 4 * 2000 * 88
    e0 = np.sqrt(ev.evaluate(start, want_residual=False)["sq_residual"] / rows)
    x, err = lm_refine(ev, start, 5, max_iter=30)
    rms = err / np.sqrt(rows)
    assert e0 > 1.0 and 0.19 < rms < 0.21
    e = po.stereo_residual(x, 5, w, l, r)
    assert np.sqrt(e @ e) == pytest.approx(err, rel=1e-9)
    assert np.abs(x[:4] - truth[:4]).max() < 0.5            # focal lengths and centre, pixels
    assert np.abs(x[-3:] - truth[-3:]).max() < 0.05         # baseline, millimetres


@pytest.mark.parametrize("views,stereo", [(2000, True), (40, True), (9, False)])
def test_refinement_loop_inside_the_library(views, stereo):
    """svcal_lm_refine (the reference's loop, CStereoCameraCalibration.cpp:278-335, as one C call)
    walks the same steps as the Python loop over svcal_solve_damped."""
    from stereovision_amd.calib import ReprojectionEvaluator, lm_refine
    w, l, r, _, start = synth.make_stereo_views(views=views, perturb=1e-2)
    a = ReprojectionEvaluator(w, l, r if stereo else None)
    b = ReprojectionEvaluator(w, l, r if stereo else None)
    start = start if stereo else np.concatenate([start[:9], start[18:18 + 6 * views]])
    # three iterations from a perturbed start: far from convergence, the same steps
    x_py, err_py = lm_refine(a, start, 5, max_iter=3)
    x_c, err_c, its = b.lm_refine(start, 5, max_iter=3)
    assert its == 3
    assert np.abs(x_c - x_py).max() < 1e-9 * max(1.0, np.abs(x_py).max())
    assert err_c == pytest.approx(err_py, rel=1e-9)
    # to the end: both arrive at the noise floor (at it, which tiny steps are still accepted
    # hangs on the last bits of the gain ratio)
    x_py, err_py = lm_refine(a, start, 5, max_iter=30)
    x_c, err_c, its = b.lm_refine(start, 5, max_iter=30)
    assert 3 < its <= 30 and err_c == pytest.approx(err_py, rel=1e-6)
    with pytest.raises(Exception):
        b.lm_refine(start, 5, max_iter=-1)


@pytest.mark.parametrize("views,stereo,k", [(6, True, 5), (6, True, 2), (7, False, 5), (2000, True, 5)])
def test_device_step_equals_host_step(views, stereo, k):
    """svcal_solve_damped against the numpy block-sparse solve of the fetched blocks (and,
    for the small problems, against the dense solve the reference performs): the same step
    and the same predicted decrease to rounding, for several dampings."""
    from stereovision_amd.calib import ReprojectionEvaluator, assemble_dense, solve_damped
    w, l, r, _, start = synth.make_stereo_views(views=views, perturb=1e-2)
    if not stereo:
        nI = 9
        start = np.concatenate([start[:nI], start[2 * nI: 2 * nI + 6 * views]])
    elif k == 2:
        start = np.concatenate([start[:6], start[9:15], start[18:]])
    ev = ReprojectionEvaluator(w, l, r if stereo else None)
    b = ev.evaluate(start, k, want_residual=False)
    assert ev.sq_residual() == b["sq_residual"]
    for lam in (0.0, 1e-3, 0.7):
        step, pred = ev.solve_damped(lam)
        want, wpred = solve_damped(b, k, views, lam, stereo)
        scale = np.abs(want).max()
        assert np.abs(step - want).max() < 1e-8 * scale
        assert pred == pytest.approx(wpred, rel=1e-8)
        if views < 100:
            A, G = assemble_dense(b, k, views, stereo)
            dense = -np.linalg.solve(A + lam * np.diag(np.diag(A)), G)
            assert np.abs(step - dense).max() < 1e-6 * scale
    ev.close()


@pytest.mark.parametrize("views,stereo,parts", [(7, True, 2), (7, False, 3), (2000, True, 4)])
def test_step_with_views_split_over_contexts(views, stereo, parts):
    """svcal_solve_begin / svcal_solve_end on several contexts of one GPU, each holding a
    range of the views, the two sums over the contexts added here as an all-reduce would:
    the assembled step and predicted decrease equal svcal_solve_damped on one context holding
    every view."""
    import torch
    from stereovision_amd.calib import (HipCalEngine, ReprojectionEvaluator, split_views, view_bounds)
    w, l, r, _, start = synth.make_stereo_views(views=views, perturb=1e-2)
    nI = 9
    if not stereo:
        start = np.concatenate([start[:nI], start[2 * nI: 2 * nI + 6 * views]])
    whole = ReprojectionEvaluator(w, l, r if stereo else None)
    whole.evaluate(start, 5, want_residual=False)
    spans = [view_bounds(views, parts, i) for i in range(parts)]
    engs = [HipCalEngine(w, l[b:e], r[b:e] if stereo else None, "cuda:0") for b, e in spans]
    for eng, (b, e) in zip(engs, spans):
        eng.set_params(split_views(start, 5, views, stereo, b, e), 5)
        eng.eval()
    total = torch.stack([eng.export_sums() for eng in engs]).sum(0)
    for eng in engs:
        eng.import_sums(total)
        assert eng.sq_residual() == pytest.approx(whole.sq_residual(), rel=1e-12)
    e0 = 2 * nI if stereo else nI
    for lam in (1e-3, 0.7):
        want, wpred = whole.solve_damped(lam)
        csum = torch.stack([eng.solve_begin(lam) for eng in engs]).sum(0)
        got = np.zeros_like(want)
        pred = 0.0
        for i, (eng, (b, e)) in enumerate(zip(engs, spans)):
            step, ps, pv = eng.solve_end(lam, csum)
            got[e0 + 6 * b:e0 + 6 * e] = step[e0:e0 + 6 * (e - b)]
            pred += pv
            if i == 0:
                got[:e0] = step[:e0]
                if stereo:
                    got[e0 + 6 * views:] = step[e0 + 6 * (e - b):]
                pred += ps
            else:
                assert np.array_equal(got[:e0], step[:e0])       # the same shared system everywhere
        assert np.abs(got - want).max() < 1e-9 * np.abs(want).max()
        assert pred == pytest.approx(wpred, rel=1e-9)
    with pytest.raises(Exception):
        engs[0].eval()
        engs[0].solve_end(1e-3, csum)             # a new evaluation: begin has to come first
    whole.close()
    for eng in engs:
        eng.ev.close()


def test_device_step_argument_errors():
    from stereovision_amd import SvicpError
    from stereovision_amd.calib import ReprojectionEvaluator
    w, l, r, _, start = synth.make_stereo_views(views=4)
    ev = ReprojectionEvaluator(w, l, r)
    with pytest.raises(SvicpError):
        ev.solve_damped(1e-3)                   # nothing evaluated yet
    with pytest.raises(SvicpError):
        ev.sq_residual()
    ev.evaluate(start, 5, want_residual=False)
    with pytest.raises(SvicpError):
        ev.solve_damped(-1.0)
    with pytest.raises(SvicpError):
        ev.solve_damped(float("nan"))
    # the split solve: end without begin, begin with a bad damping, NULL buffers
    import torch
    buf = torch.zeros(90, dtype=torch.float64, device="cuda:0")
    with pytest.raises(SvicpError):
        ev.solve_end(1e-3, buf.data_ptr())
    with pytest.raises(SvicpError):
        ev.solve_begin(-1.0, buf.data_ptr())
    with pytest.raises(SvicpError):
        ev.solve_begin(1e-3, 0)
    ev.solve_begin(1e-3, buf.data_ptr())
    with pytest.raises(SvicpError):
        ev.solve_end(1e-3, 0)
    step, ps, pv = ev.solve_end(1e-3, buf.data_ptr())
    want, wpred = ev.solve_damped(1e-3)
    assert np.array_equal(step, want) and ps + pv == wpred     # one context: the same kernels, the same sums
    with pytest.raises(SvicpError):
        ev.solve_end(1e-3, buf.data_ptr())                      # consumed: begin again first
    ev.close()


def test_dense_and_block_sparse_loops_agree(po):
    from stereovision_amd.calib import ReprojectionEvaluator, lm_refine
    w, l, r, _, start = synth.make_stereo_views(views=6)
    ev = ReprojectionEvaluator(w, l, r)
    xa, ea = lm_refine(ev, start, 5, max_iter=15, dense=True)
    xh, eh = lm_refine(ev, start, 5, max_iter=15, host_solve=True)
    assert ea == pytest.approx(eh, rel=1e-6)
    xb, eb = lm_refine(ev, start, 5, max_iter=15, dense=False)
    # the two solves differ in rounding only, but the system is ill conditioned (k3 is barely
    # observable), so the iterates drift apart in the eighth digit of the error
    assert ea == pytest.approx(eb, rel=1e-6)
    assert np.abs(xa[:4] - xb[:4]).max() < 1e-2 and np.abs(xa[-3:] - xb[-3:]).max() < 1e-2


# ---- doStereoRectify in the library ---------------------------------------------
def test_stereo_rectify_in_the_library(po):
    """svcal_stereo_rectify (CStereoCameraRectification.cpp:13-58 as a C call) against the oracle
    and the numpy version: the usual rig, no rotation at all, a large rotation, and a rotation
    matrix that is slightly off orthogonal (the reference's SVD takes the nearest rotation)."""
    from scipy.spatial.transform import Rotation
    from stereovision_amd.calib import stereo_rectify, stereo_rectify_native
    KL = np.array([[535.0, 0, 322.0], [0, 537.0, 243.0], [0, 0, 1]])
    KR = np.array([[533.0, 0, 318.0], [0, 534.0, 239.0], [0, 0, 1]])
    T = np.array([-90.0, 1.0, 2.0])
    for rv in ([0.01, -0.02, 0.005], [0.0, 0.0, 0.0], [0.4, -0.9, 0.6], [2.5, 0.3, -1.0]):
        R = Rotation.from_rotvec(rv).as_matrix()
        lp, rp, Q = stereo_rectify_native(KL, KR, R, T)
        ok, pl, pr, oQ = po.stereo_rectify(KL, KR, R, T)
        assert ok and np.allclose(lp, pl, atol=1e-9) and np.allclose(rp, pr, atol=1e-9) and np.allclose(Q, oQ, atol=1e-12)
        npl, npr, nQ = stereo_rectify(KL, KR, R, T)
        assert np.allclose(lp, npl, atol=1e-9) and np.allclose(rp, npr, atol=1e-9) and np.allclose(Q, nQ, atol=1e-12)
    R = Rotation.from_rotvec([0.01, -0.02, 0.005]).as_matrix()
    off = R + 1e-7 * np.arange(9.0).reshape(3, 3)
    lp, rp, Q = stereo_rectify_native(KL, KR, off, T)
    lp0, rp0, Q0 = stereo_rectify_native(KL, KR, R, T)
    assert np.abs(lp - lp0).max() < 1e-3 and np.abs(rp - rp0).max() < 1e-3
    with pytest.raises(Exception):
        stereo_rectify_native(KL, KR, np.zeros((3, 3)), T)          # no inverse
    with pytest.raises(Exception):
        stereo_rectify_native(KL, KR, np.eye(3), np.array([0.0, 0.0, 5.0]))   # no x / y baseline


# ---- rectification maps (CStereoCameraRectification::doImageRectify) -----------
@pytest.mark.parametrize("size,k", [((640, 480), 5), ((640, 480), 2), ((1921, 1083), 5)])
def test_rectify_maps_match_oracle(po, size, k):
    """Same operations in fp64 on both sides; the inverse of the projection is formed by the
    same elimination.  Tolerance 1e-9 pixel; the in/out-of-image decision may flip only for
    samples within that distance of the border."""
    from scipy.spatial.transform import Rotation
    from stereovision_amd.calib import rectify_maps, stereo_rectify
    KL = np.array([[535.0, 0.3, 322.0], [0, 537.0, 243.0], [0, 0, 1]])   # with a little skew
    KR = np.array([[533.0, 0, 318.0], [0, 534.0, 239.0], [0, 0, 1]])
    dL = np.array([-0.28, 0.09, 1e-3, -5e-4, 0.01])[:k]
    dR = np.array([-0.27, 0.08, -7e-4, 4e-4, -0.02])[:k]
    R = Rotation.from_rotvec([0.01, -0.02, 0.005]).as_matrix()
    pl, pr, _ = stereo_rectify(KL, KR, R, [-90.0, 1.0, 2.0])
    w, h = size
    got = rectify_maps(w, h, KL, dL, KR, dR, pl, pr)
    ok, want = po.rectify_maps(w, h, KL, dL, KR, dR, pl, pr)
    assert ok
    for g, e in zip(got, want):
        differ = (g == 9.99e4) != (e == 9.99e4)
        assert differ.sum() <= 4
        assert np.abs(g - e)[~differ].max() < 1e-9
    assert (got[0] != 9.99e4).mean() > 0.5


def test_rectify_maps_argument_errors():
    from stereovision_amd import SvicpError
    from stereovision_amd.calib import rectify_maps
    K = np.eye(3)
    with pytest.raises(SvicpError):
        rectify_maps(0, 10, K, np.zeros(5), K, np.zeros(5), K, K)
    with pytest.raises(SvicpError):
        rectify_maps(10, 10, K, np.zeros(3), K, np.zeros(3), K, K)
    with pytest.raises(SvicpError):
        rectify_maps(10, 10, K, np.zeros(5), K, np.zeros(5), np.zeros((3, 3)), K)


def test_fixed_intrinsics_step_and_refinement_on_reference_corners(po):
    """update_intrinsics = false (CStereoCameraCalibration.cpp:187-245, :284-285, :306) on the
    reference's own corner tables: the device step leaves the intrinsics alone and equals the
    dense solve of the oracle's restated Jacobian; the loop then refines poses only."""
    from stereovision_amd.calib import ReprojectionEvaluator, lm_refine, solve_damped
    c = np.load(os.path.join(GOLD, "testdata_corners.npz"))
    w, l, r = c["world_stereo"], c["left"], c["right"]
    V, k, off = 4, 5, 18
    x0 = stereo_start(w, l, r, (640, 480), 535.0)
    ev = ReprojectionEvaluator(w, l, r)
    ev.set_params(x0, k)
    ev.eval()
    full_step, _ = ev.solve_damped(1e-3)
    ev.set_update_intrinsics(False)
    step, pred = ev.solve_damped(1e-3)
    assert np.all(step[:off] == 0.0) and np.abs(full_step[:off]).max() > 0
    J = po.stereo_jacobian_dense(x0, k, w, V, update_intrinsics=False)
    e = po.stereo_residual(x0, k, w, l, r)
    A, G = J.T @ J, J.T @ e
    dense = -np.linalg.solve(A + 1e-3 * np.diag(np.diag(A)), G)
    assert np.abs(step[off:] - dense).max() <= 1e-5 * np.abs(dense).max()
    assert pred == pytest.approx(float(dense @ (1e-3 * dense - G)), rel=1e-5)
    host_step, host_pred = solve_damped(ev.fetch(False), k, V, 1e-3, update_intrinsics=False)
    assert np.abs(step - host_step).max() <= 1e-9 * np.abs(host_step).max()
    ev.set_update_intrinsics(True)
    again, _ = ev.solve_damped(1e-3)
    assert np.array_equal(again, full_step)
    # the loop: same result on the device, block-sparse on the host and densely
    x_dev, err_dev = lm_refine(ev, x0, k, max_iter=30, update_intrinsics=False)
    x_host, err_host = lm_refine(ev, x0, k, max_iter=30, host_solve=True, update_intrinsics=False)
    x_dense, err_dense = lm_refine(ev, x0, k, max_iter=30, dense=True, update_intrinsics=False)
    assert np.array_equal(x_dev[:off], x0[:off]) and np.array_equal(x_host[:off], x0[:off])
    assert err_dev == pytest.approx(err_host, rel=1e-6) and err_dev == pytest.approx(err_dense, rel=1e-6)
    e0 = po.stereo_residual(x0, k, w, l, r)
    assert err_dev < np.sqrt(e0 @ e0)
    ev.close()
