"""GPU parity of the point-to-point ICP (svicp_run_point2point) against the CPU oracle.

Tolerances: correspondences and selected counts bit-exact; per-iteration rotation 1e-9 and
translation 1e-9 x extent against the oracle's fp64-sum mode (same terms, different summation
order only); 1e-4 / 1e-6 rad against the oracle's faithful mode, whose fp32 running sums are
order-dependent by themselves.
"""
import os
import subprocess

import numpy as np
import pytest

import refmath

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ctx(ref, src, **opt):
    import stereovision_amd as sv
    c = sv.Context(0, dist_threshold=15.0, **opt)
    c.set_target(ref, ref)             # no normals on this path
    c.set_source(src)
    return c


@pytest.mark.parametrize("graph", [1, 0])
def test_bundled_pair_matches_oracle(po, pair, graph):
    ref, src, _ = pair
    c = _ctx(ref, src, use_graph=graph)
    T, info = c.run_point2point(5)
    RT, I = c.history_rt()
    wide = po.icp_point2point(po.KDTree(ref), src, 5, faithful=False)
    extent = np.ptp(ref, axis=0).max()
    assert np.array_equal(I[:, 0], wide["info"][:, 0])            # selected counts
    assert np.allclose(I[:, 3], wide["info"][:, 1], rtol=1e-12)   # weight sums
    for k in range(5):
        assert np.abs(RT[k, :9] - wide["RT"][k, :9]).max() < 1e-9     # rotation entries
        assert np.abs(RT[k, 9:] - wide["RT"][k, 9:]).max() < 1e-9 * extent
    assert np.abs(T - wide["trans"]).max() < 1e-9 * extent
    assert info.inliers == len(src) and info.iterations == 5 and info.status == 0
    assert np.abs(c.get_source() - wide["src"]).max() <= 4e-5
    faithful = po.icp_point2point(po.KDTree(ref), src, 5, faithful=True)
    assert np.abs(T - faithful["trans"]).max() < 1e-4
    assert refmath.rot_angle(T[:3, :3], faithful["trans"][:3, :3]) < 1e-6


def test_synthetic_pair_and_second_run(po):
    from stereovision_amd import synth
    tgt, _, src, pose = synth.make_pair(96)
    c = _ctx(tgt, src)
    T, info = c.run_point2point(12)
    wide = po.icp_point2point(po.KDTree(tgt), src, 12, faithful=False, history=False)
    extent = np.ptp(tgt, axis=0).max()
    assert np.abs(T[:3, :3] - wide["trans"][:3, :3]).max() < 1e-8
    assert np.abs(T - wide["trans"]).max() < 1e-8 * extent
    assert info.inliers == wide["info"][-1, 0]
    # a second run from the same start replays the graph and gives the same answer
    c.set_source(src)
    T2, _ = c.run_point2point(12)
    assert np.abs(T2 - T).max() < 1e-10
    # after a point-to-plane run the point-to-point history is gone
    import stereovision_amd as sv
    c.set_target(tgt)
    c.set_source(src)
    c.run(3)
    with pytest.raises(sv.SvicpError):
        c.history_rt()


def test_nothing_selected_is_reported(po):
    import stereovision_amd as sv
    from stereovision_amd import synth
    tgt, _, src, _ = synth.make_pair(32)
    c = _ctx(tgt, src + np.float32(500.0))
    with pytest.raises(sv.SvicpError) as e:
        c.run_point2point(2)
    assert e.value.status == sv._lib.ERR_SINGULAR


def test_python_and_cpp_drop_in_classes(po, pair, tmp_path, capsys):
    import stereovision_amd as sv
    ref, src, _ = pair
    wide = po.icp_point2point(po.KDTree(ref), src, 5, faithful=False, history=False)
    icp = sv.CPoint2PointICP(verbose=True)
    icp.setReferenceData(ref)
    move = src.copy()
    icp.setMoveData(move)
    icp.setEpsilon(1e-6)
    icp.setMaxIterateCount(5)
    icp.run()
    assert "Current Transformation:" in capsys.readouterr().out
    assert np.abs(icp.getTransform() - wide["trans"]).max() < 1e-6
    assert np.abs(move - wide["src"]).max() <= 4e-5              # updated in place, 4 times
    exe = os.path.join(ROOT, "examples", "test_icp")
    np.savetxt(tmp_path / "17.txt", ref, fmt="%.8f")
    np.savetxt(tmp_path / "15.txt", src, fmt="%.8f")
    out = subprocess.run([exe, str(tmp_path / "17.txt"), str(tmp_path / "15.txt"), "p2p"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.splitlines()
    at = lines.index("Current Transformation:")
    Tc = np.array([[float(v) for v in lines[at + 1 + r].split()] for r in range(4)])
    assert np.abs(Tc - wide["trans"]).max() < 1e-5               # printed with 6 digits


def test_full_size_against_oracle(po):
    """1M points, three iterations: selected counts equal, increments equal to the oracle's
    fp64-sum mode; the selected count grows as the clouds come together."""
    from stereovision_amd import synth
    tgt, _, src, pose = synth.make_pair(1024)
    c = _ctx(tgt, src)
    T, info = c.run_point2point(3)
    RT, I = c.history_rt()
    wide = po.icp_point2point(po.KDTree(tgt), src, 3, faithful=False, threads=po.max_threads(),
                              history=False)
    extent = np.ptp(tgt, axis=0).max()
    assert np.array_equal(I[:, 0], wide["info"][:, 0])
    assert I[2, 0] > I[1, 0] > I[0, 0] > 0.25 * len(src)
    assert np.abs(RT[:, :9] - wide["RT"][:, :9]).max() < 1e-9
    assert np.abs(RT[:, 9:] - wide["RT"][:, 9:]).max() < 1e-9 * extent
    assert np.abs(T - wide["trans"]).max() < 1e-9 * extent
