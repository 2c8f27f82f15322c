"""Deterministic synthetic cloud pairs for the benchmark configurations.

Counter-based generator: value i of stream `seed` is splitmix64(seed + i), so
every consumer (CPU checker, GPU path, any rank) sees identical bits without
sharing state.  The surface mimics the density of the reference's bundled
pair (PointCloud/Registration/15.txt, 17.txt: median neighbour spacing 2.46).
"""
import numpy as np

SEED_TARGET = 0x5EED0001
SEED_SOURCE = 0x5EED0002
SEED_NOISE = 0x5EED0003

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(seed, start, count):
    """count doubles in [0, 1) from stream `seed`, positions start .. start+count-1."""
    with np.errstate(over="ignore"):
        ctr = np.uint64(seed) * np.uint64(0x100000001B3) + np.arange(start, start + count, dtype=np.uint64)
    return (splitmix64(ctr) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def _height(x, y):
    return 40.0 * np.sin(x / 180.0) * np.cos(y / 230.0) + 15.0 * np.sin(x / 37.0 + y / 53.0)


def _normal(x, y):
    dzdx = 40.0 / 180.0 * np.cos(x / 180.0) * np.cos(y / 230.0) + 15.0 / 37.0 * np.cos(x / 37.0 + y / 53.0)
    dzdy = -40.0 / 230.0 * np.sin(x / 180.0) * np.sin(y / 230.0) + 15.0 / 53.0 * np.cos(x / 37.0 + y / 53.0)
    n = np.stack([-dzdx, -dzdy, np.ones_like(dzdx)], axis=1)
    return n / np.linalg.norm(n, axis=1, keepdims=True)


OFFSET = np.array([-60.0, -40.0, -160.0])


def _surface(side, seed, pair):
    n = side * side
    base = pair * 4 * n
    u = uniform01(seed, base, n)
    v = uniform01(seed, base + n, n)
    i = np.repeat(np.arange(side, dtype=np.float64), side)
    j = np.tile(np.arange(side, dtype=np.float64), side)
    x = (i + u) * 2.5
    y = (j + v) * 2.5
    return x, y


def rot_zyx(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def make_pair(side, pair=0, pose=None):
    """Returns (target (M,3) f32, normals (M,3) f32, source (M,3) f32, pose 4x4 f64).

    target: jittered-grid height field with analytic unit normals.
    source: the same surface re-jittered, + N(0, 0.05^2) noise, moved by `pose`
    about the cloud centroid (default Rz(0.6 deg) Ry(-0.4 deg) Rx(0.5 deg),
    t = (-0.4, 1.2, 0.8)).  `pair` selects an independent pair of streams.
    """
    n = side * side
    x, y = _surface(side, SEED_TARGET, pair)
    tgt = np.stack([x, y, _height(x, y)], axis=1) + OFFSET
    nrm = _normal(x, y)

    xs, ys = _surface(side, SEED_SOURCE, pair)
    src = np.stack([xs, ys, _height(xs, ys)], axis=1) + OFFSET
    base = pair * 8 * n
    u1 = uniform01(SEED_NOISE, base, 3 * n).reshape(n, 3)
    u2 = uniform01(SEED_NOISE, base + 3 * n, 3 * n).reshape(n, 3)
    gauss = np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)
    src = src + 0.05 * gauss
    if pose is None:
        d = np.pi / 180.0
        R = rot_zyx(0.5 * d, -0.4 * d, 0.6 * d)
        t = np.array([-0.4, 1.2, 0.8])
    else:
        R, t = pose[:3, :3], pose[:3, 3]
    c = tgt.mean(axis=0)
    src = (src - c) @ R.T + c + t
    P = np.eye(4)
    P[:3, :3] = R
    P[:3, 3] = c + t - R @ c
    return tgt.astype(np.float32), nrm.astype(np.float32), src.astype(np.float32), P
