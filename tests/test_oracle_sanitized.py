"""The CPU oracle under AddressSanitizer and UBSan: oracle/asan_driver.c links the three oracle
sources directly (a program of its own, not a preloaded library) and runs calcNormals, five
iterations of the point-to-plane loop on the reference's bundled pair (as test.cpp:23), three
of the point-to-point loop, one stereo evaluation on the reference's corner tables and the text
parser, then frees everything.  The restatement follows code that has a delete / new[] mismatch
(CPoint2PlaneICP.cpp:129,153), leaks (:16, CTxtReader.cpp:52) and members read before they are
set (:3-6); none of that may have come along.  The driver's numbers must equal the goldens the
unsanitized build produced."""
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.mark.skipif(shutil.which("gcc") is None and shutil.which("cc") is None, reason="needs a C compiler")
def test_oracle_runs_clean_under_address_and_undefined_behaviour_sanitizers(tmp_path, pair, po):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan_driver"],
                          stdout=subprocess.DEVNULL)
    ref, src, gold = pair
    ref.astype(np.float32).tofile(tmp_path / "ref.f32")
    src.astype(np.float32).tofile(tmp_path / "src.f32")
    c = np.load(os.path.join(GOLD, "testdata_corners.npz"))
    V, C, K = 4, 54, 5
    intr = [535.0, 535.0, 320.0, 240.0]
    dist = [-0.1, 0.02, 1e-3, -5e-4, 1e-3]
    views = [[0.05 * (v - 1.5), -0.04 * v, 0.02, -100.0, -60.0, 500.0 + 20.0 * v] for v in range(V)]
    vec = np.array(intr + dist + intr + dist + sum(views, []) + [0.01, -0.02, 0.005, -90.0, 1.0, 2.0])
    assert len(vec) == 2 * (4 + K) + 6 * V + 6
    np.concatenate([vec, c["world_stereo"].ravel(), c["left"].ravel(), c["right"].ravel()]).astype(
        np.float64).tofile(tmp_path / "stereo.f64")
    env = dict(os.environ)
    env.pop("LD_PRELOAD", None)            # (the sanitizer runtime must come first in the process)
    # leak detection needs ptrace rights some containers withhold: asked for, and dropped only
    # if the sanitizer itself says it cannot run it
    for leaks in ("1", "0"):
        env["ASAN_OPTIONS"] = f"detect_leaks={leaks}:abort_on_error=0:halt_on_error=1"
        env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
        r = subprocess.run([os.path.join(ROOT, "oracle", "asan_driver"), str(tmp_path / "ref.f32"), str(len(ref)),
                            str(tmp_path / "src.f32"), str(len(src)), str(tmp_path / "stereo.f64"),
                            str(V), str(C), str(K), str(tmp_path / "out.f64")],
                           env=env, capture_output=True, text=True, timeout=600)
        if leaks == "1" and "LeakSanitizer has encountered a fatal error" in r.stderr:
            continue
        break
    assert r.returncode == 0, r.stderr[-3000:]
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-3000:]
    assert "asan_driver: done, status 0" in r.stdout
    out = np.fromfile(tmp_path / "out.f64")
    # -O1 with sanitizers against the -O2 build behind the goldens: the same IEEE operations
    # (contraction is off in both), so the same numbers
    assert np.array_equal(out[:16].reshape(4, 4), gold["trans5"])
    assert np.array_equal(out[16:46].reshape(5, 6), gold["X5"])
    want = po.stereo_eval_sparse(vec, K, c["world_stereo"], c["left"], c["right"])
    assert out[46] == want["sq_residual"]
    assert out[47] == pytest.approx(float(np.abs(gold["normals"]).sum(dtype=np.float64)), rel=1e-6)
    tree = po.KDTree(ref)
    p2p = po.icp_point2point(tree, src, 3, faithful=True, threads=1, history=False)
    assert np.array_equal(out[48:64].reshape(4, 4), p2p["trans"])
