"""CPU tests of the boundary: the shared library loads, exports every symbol the
header declares, and refuses to work without a GPU instead of falling back."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols(header="svicp.h", prefix="svicp_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from stereovision_amd import _lib
    lib = _lib.load()
    declared = _header_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in svicp.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "ctypes table and header disagree"


def test_calibration_symbols_are_exported():
    from stereovision_amd import _lib, calib
    lib = _lib.load()
    declared = _header_symbols("svcal.h", "svcal_")
    assert len(declared) == 19
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in svcal.h but not exported"
    assert sorted(calib.CAL_SYMBOLS) == declared
    assert ctypes.sizeof(calib._Blocks) == 12 * 8


def test_calibration_needs_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import stereovision_amd as sv
    from stereovision_amd import calib
    with pytest.raises(sv.SvicpError) as e:
        calib.ReprojectionEvaluator(np.zeros((4, 2)), np.zeros((1, 4, 2)))
    assert e.value.status == sv._lib.ERR_NO_DEVICE


def test_struct_layouts_match_header():
    from stereovision_amd import _lib
    assert ctypes.sizeof(_lib.Options) == 28
    assert ctypes.sizeof(_lib.Info) == 40
    o = _lib.Options()
    assert _lib.load().svicp_default_options(ctypes.byref(o)) == 0
    assert o.dist_threshold == 5.0 and o.normals_k == 20        # CPoint2PlaneICP.cpp:68,125
    assert o.use_graph == 1 and o.sort_source == 1 and o.coherent == 1 and o.centred_normals == 0


def test_no_device_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import stereovision_amd as sv
    with pytest.raises(sv.SvicpError) as e:
        sv.Context(0)
    assert e.value.status == sv._lib.ERR_NO_DEVICE
    with pytest.raises(sv.SvicpError):
        sv.CPoint2PlaneICP()


def test_null_arguments_are_rejected():
    from stereovision_amd import _lib
    lib = _lib.load()
    assert lib.svicp_create(0, None) == _lib.ERR_ARG
    assert lib.svicp_default_options(None) == _lib.ERR_ARG
    assert lib.svicp_destroy(None) == _lib.OK
    assert lib.svicp_last_error(None) is not None


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "stereovision_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), f"{f} mentions the oracle"


def test_move_data_must_be_mutable_float32():
    import stereovision_amd as sv
    icp = sv.CPoint2PlaneICP.__new__(sv.CPoint2PlaneICP)   # no context: argument checks only
    icp._move = None
    with pytest.raises(ValueError):
        sv.CPoint2PlaneICP.setMoveData(icp, np.zeros((4, 3)))          # float64
    with pytest.raises(ValueError):
        sv.CPoint2PlaneICP.setMoveData(icp, np.zeros((4, 2), np.float32))
    sv.CPoint2PlaneICP.setMoveData(icp, np.zeros((4, 3), np.float32))
