"""MI355X-native point-to-plane ICP registration (gfx950 HIP kernels behind a C ABI).

Python mirror of the reference's registration classes:
CPoint2PlaneICP, CKNNSearch, CTxtReader (PointCloud/Registration/*.h).
"""
from ._lib import Info, Options, SvicpError, load  # noqa: F401
from .icp import (ArenaBatch, Batch, CKNNSearch, CPoint2PlaneICP, CPoint2PointICP, CTxtReader,  # noqa: F401
                  Context)  # noqa: F401

__all__ = [
    "ArenaBatch", "Batch", "CKNNSearch", "CPoint2PlaneICP", "CPoint2PointICP", "CTxtReader", "Context", "Info", "Options",
    "SvicpError", "load",
]
