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


# measured on this surface (sides 256 .. 2048): share of the source with a target point within
# squared distance 5 at the start, against the factor k applied to the default pose (angles
# k * 256 / side of the default ones, shift k of the default one)
_INLIER_TABLE = ((0.25, 0.97), (0.5, 0.85), (0.6, 0.78), (0.7, 0.69), (0.8, 0.63), (1.0, 0.55))


def pose_for_inliers(side, inlier_frac=0.7):
    """A start pose that leaves about `inlier_frac` (0.56 .. 0.96) of the source within the
    weight function's threshold of the target, whatever the cloud's extent: the default
    angles scaled with 256 / side (a rotation's displacement grows with the distance from
    the centroid, i.e. with the extent), angles and shift both scaled by a factor read off
    the measured table, 7 % on the safe side."""
    fr = float(np.clip(inlier_frac, 0.56, 0.96))
    ks = np.array([k for k, _ in _INLIER_TABLE])
    fs = np.array([f for _, f in _INLIER_TABLE])
    k = 0.93 * float(np.interp(fr, fs[::-1], ks[::-1]))
    d = np.pi / 180.0
    a = k * 256.0 / side
    P = np.eye(4)
    P[:3, :3] = rot_zyx(0.5 * d * a, -0.4 * d * a, 0.6 * d * a)
    P[:3, 3] = k * np.array([-0.4, 1.2, 0.8])
    return P


def make_pair(side, pair=0, pose=None, inlier_frac=None):
    """Returns (target (M,3) f32, normals (M,3) f32, source (M,3) f32, pose 4x4 f64).

    target: jittered-grid height field with analytic unit normals.
    source: the same surface re-jittered, + N(0, 0.05^2) noise, moved by `pose`
    about the cloud centroid (default Rz(0.6 deg) Ry(-0.4 deg) Rx(0.5 deg),
    t = (-0.4, 1.2, 0.8); with `inlier_frac` and no pose: pose_for_inliers).
    `pair` selects an independent pair of streams.
    """
    if pose is None and inlier_frac is not None:
        pose = pose_for_inliers(side, inlier_frac)
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


SEED_POSE = 0x5EED0004


def small_pose(pair):
    """A per-pair random small pose: angles within +-0.8 degrees, shift within +-1.2."""
    u = uniform01(SEED_POSE, 6 * pair, 6) * 2.0 - 1.0
    d = np.pi / 180.0
    P = np.eye(4)
    P[:3, :3] = rot_zyx(0.8 * d * u[0], 0.8 * d * u[1], 0.8 * d * u[2])
    P[:3, 3] = 1.2 * u[3:]
    return P


SEED_CALIB = 0x5EED0005


def rodrigues(r):
    """Rotation vector -> matrix, I + sin(t) K + (1 - cos(t)) K^2."""
    r = np.asarray(r, np.float64)
    t = np.linalg.norm(r)
    if t < 1e-6:
        return np.eye(3)
    a = r / t
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(t) * K + (1 - np.cos(t)) * (K @ K)


def project(intr, dist, E, world_xy):
    """The reference's five-term projection model (r = nx^2 + ny^2)."""
    p = np.concatenate([world_xy, np.zeros((len(world_xy), 1)), np.ones((len(world_xy), 1))], axis=1)
    c = p @ E.T
    nx, ny = c[:, 0] / c[:, 2], c[:, 1] / c[:, 2]
    r = nx * nx + ny * ny
    k1, k2, p1, p2, k3 = dist
    rad = 1.0 + k1 * r + k2 * r * r + k3 * r * r * r
    xe = nx * rad + 2.0 * p1 * nx * ny + p2 * (r * r + 2.0 * nx * nx)
    ye = ny * rad + p1 * (r * r + 2.0 * ny * ny) + 2.0 * p2 * nx * ny
    return np.stack([intr[0] * xe + intr[2], intr[1] * ye + intr[3]], axis=1)


def make_stereo_views(views=2000, board=(11, 8), square=25.0, noise=0.2, perturb=1e-3):
    """Synthetic stereo calibration problem: an 11 x 8 board at 25 mm seen in `views`
    poses by two cameras.  Returns (world (C,2), left (V,C,2), right (V,C,2),
    true parameter vector, perturbed start vector) in the stereo layout with five
    distortion terms: [intrL(4) distL(5) | intrR(4) distR(5) | V x (rvec, t) | rvecS tS]."""
    bx, by = board
    gx, gy = np.meshgrid(np.arange(bx) * square, np.arange(by) * square)
    world = np.stack([gx.ravel(), gy.ravel()], axis=1).astype(np.float64)
    C = len(world)
    u = uniform01(SEED_CALIB, 0, 6 * views).reshape(views, 6)
    rvec = (u[:, :3] - 0.5) * 0.7
    t = np.stack([-120 + 100 * u[:, 3], -100 + 100 * u[:, 4], 450 + 450 * u[:, 5]], axis=1)
    intr = np.array([535.0, 535.0, 320.0, 240.0])
    dist = np.array([-0.28, 0.09, 1e-3, -5e-4, 0.0])
    pose = np.array([0.01, -0.02, 0.005, -90.0, 1.0, 2.0])
    S = np.concatenate([rodrigues(pose[:3]), pose[3:, None]], axis=1)
    g1 = uniform01(SEED_CALIB, 6 * views, 4 * views * C).reshape(2, views, C, 2)
    g2 = uniform01(SEED_CALIB, 6 * views + 4 * views * C, 4 * views * C).reshape(2, views, C, 2)
    gauss = np.sqrt(-2.0 * np.log(1.0 - g1)) * np.cos(2.0 * np.pi * g2)
    left = np.empty((views, C, 2))
    right = np.empty((views, C, 2))
    for v in range(views):
        E = np.concatenate([rodrigues(rvec[v]), t[v][:, None]], axis=1)
        Er = S @ np.concatenate([E, [[0, 0, 0, 1]]], axis=0)
        left[v] = project(intr, dist, E, world) + noise * gauss[0, v]
        right[v] = project(intr, dist, Er, world) + noise * gauss[1, v]
    truth = np.concatenate([intr, dist, intr, dist, np.concatenate([rvec, t], axis=1).ravel(), pose])
    w = uniform01(SEED_CALIB + 1, 0, len(truth)) - 0.5
    start = truth * (1.0 + perturb * w) + perturb * w
    return world, left, right, truth, start
