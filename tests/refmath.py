"""Independent numpy restatement of one ICP solve, used to cross-check the C
oracle (and as the engine of the CPU-only sharding test).  fp32 where the
reference computes in float, fp64 elsewhere (CPoint2PlaneICP.cpp:68-119)."""
import numpy as np

F = np.float32


def weights(d2, thr=5.0):
    d2 = d2.astype(F)
    sl = (d2 / F(thr)).astype(F)
    ss = (sl * sl).astype(F)
    w = ((1.0 - ss.astype(np.float64)) * (1.0 - ss.astype(np.float64))).astype(F)
    return np.where(d2 <= F(thr), w, F(0))


def rows(src, tgt, nrm, idx):
    s = src.astype(F)
    n = nrm[idx].astype(F)
    d = tgt[idx].astype(F)
    a0 = (s[:, 1] * n[:, 2]).astype(F) - (s[:, 2] * n[:, 1]).astype(F)
    a1 = (s[:, 2] * n[:, 0]).astype(F) - (s[:, 0] * n[:, 2]).astype(F)
    a2 = (s[:, 0] * n[:, 1]).astype(F) - (s[:, 1] * n[:, 0]).astype(F)
    b = (((d[:, 0] - s[:, 0]).astype(F) * n[:, 0]).astype(F)
         + ((d[:, 1] - s[:, 1]).astype(F) * n[:, 1]).astype(F)).astype(F)
    b = (b + ((d[:, 2] - s[:, 2]).astype(F) * n[:, 2]).astype(F)).astype(F)
    A = np.stack([a0, a1, a2, n[:, 0], n[:, 1], n[:, 2]], axis=1).astype(np.float64)
    return A, b.astype(np.float64)


def sums32(src, tgt, nrm, idx, d2, thr=5.0):
    """The 32-double block a shard contributes: 21 upper-triangle JtJ, 6 Jtr,
    inliers, sum w b^2, sum w, 2 pad."""
    w = weights(d2, thr).astype(np.float64)
    A, b = rows(src, tgt, nrm, np.where(idx >= 0, idx, 0))
    w = np.where(idx >= 0, w, 0.0)
    JtJ = (A * w[:, None]).T @ A
    Jtr = (A * w[:, None]).T @ b
    out = np.zeros(32)
    v = 0
    for r in range(6):
        for c in range(r, 6):
            out[v] = JtJ[r, c]
            v += 1
    out[21:27] = Jtr
    out[27] = np.count_nonzero(w > 0)
    out[28] = np.sum(w * b * b)
    out[29] = np.sum(w)
    return out


def solve_from_sums(S):
    JtJ = np.zeros((6, 6))
    v = 0
    for r in range(6):
        for c in range(r, 6):
            JtJ[r, c] = JtJ[c, r] = S[v]
            v += 1
    return np.linalg.solve(JtJ, S[21:27])


def euler_to_RT(X):
    cx, sx = np.cos(X[0]), np.sin(X[0])
    cy, sy = np.cos(X[1]), np.sin(X[1])
    cz, sz = np.cos(X[2]), np.sin(X[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = Rz @ Ry @ Rx
    M[:3, 3] = X[3:6]
    return M


def apply_update(src, M):
    """p <- float(R p + T), fp64 math left to right (CPoint2PlaneICP.cpp:44-49)."""
    p = src.astype(np.float64)
    R, T = M[:3, :3], M[:3, 3]
    out = np.empty_like(p)
    for r in range(3):
        out[:, r] = ((R[r, 0] * p[:, 0] + R[r, 1] * p[:, 1]) + R[r, 2] * p[:, 2]) + T[r]
    return out.astype(F)


def rot_angle(Ra, Rb):
    """Angle of Ra^T Rb in radians."""
    c = (np.trace(Ra.T @ Rb) - 1.0) / 2.0
    return float(np.arccos(np.clip(c, -1.0, 1.0)))


def _homography(world, img):
    """DLT homography world (C,2) -> img (C,2), normalised coordinates."""
    def norm(p):
        c = p.mean(0)
        s = np.sqrt(2.0) / np.sqrt(((p - c) ** 2).sum(1)).mean()
        T = np.array([[s, 0, -s * c[0]], [0, s, -s * c[1]], [0, 0, 1]])
        return (p - c) * s, T
    w, Tw = norm(world)
    m, Tm = norm(img)
    rows = []
    for (X, Y), (u, v) in zip(w, m):
        rows.append([X, Y, 1, 0, 0, 0, -u * X, -u * Y, -u])
        rows.append([0, 0, 0, X, Y, 1, -v * X, -v * Y, -v])
    _, _, vt = np.linalg.svd(np.asarray(rows))
    H = np.linalg.inv(Tm) @ vt[-1].reshape(3, 3) @ Tw
    return H / H[2, 2]


def rotvec(R):
    """Rotation matrix -> rotation vector."""
    from scipy.spatial.transform import Rotation
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    if np.linalg.det(R) < 0:
        R = -R
    return Rotation.from_matrix(R).as_rotvec()


def homography_start(world, img, size, focal=None):
    """A starting vector [fx fy cx cy 0 0 0 0 0 | V x (rvec, t)] for a planar board: the
    principal point at the image centre, one focal length from the orthogonality constraint
    of the homographies' first two columns, extrinsics from K^-1 H."""
    cx, cy = size[0] / 2.0, size[1] / 2.0
    Hs = [_homography(world, im) for im in img]
    if focal is None:
        est = []
        for H in Hs:
            Hc = np.array([[1, 0, -cx], [0, 1, -cy], [0, 0, 1.0]]) @ H
            h1, h2 = Hc[:, 0], Hc[:, 1]
            for num, den in ((-(h1[0] * h2[0] + h1[1] * h2[1]), h1[2] * h2[2]),
                             (-(h1[0] ** 2 + h1[1] ** 2 - h2[0] ** 2 - h2[1] ** 2),
                              h1[2] ** 2 - h2[2] ** 2)):
                if abs(den) > 1e-12 and num / den > 0:
                    est.append(np.sqrt(num / den))
        focal = float(np.median(est))
    K = np.array([[focal, 0, cx], [0, focal, cy], [0, 0, 1.0]])
    out = [focal, focal, cx, cy, 0, 0, 0, 0, 0]
    for H in Hs:
        M = np.linalg.inv(K) @ H
        lam = 2.0 / (np.linalg.norm(M[:, 0]) + np.linalg.norm(M[:, 1]))
        if M[2, 2] * lam < 0:
            lam = -lam
        r1, r2, t = M[:, 0] * lam, M[:, 1] * lam, M[:, 2] * lam
        R = np.stack([r1, r2, np.cross(r1, r2)], axis=1)
        out += list(rotvec(R)) + list(t)
    return np.asarray(out, np.float64)


def stereo_start(world, left, right, size, focal=None):
    """[intrL distL | intrR distR | V x left (rvec, t) | pose] from two monocular starts;
    the pose is the median over views of right extrinsic * inverse(left extrinsic)."""
    from scipy.spatial.transform import Rotation
    a = homography_start(world, left, size, focal)
    b = homography_start(world, right, size, focal)
    V = len(left)
    rv, tv = [], []
    for v in range(V):
        Rl = Rotation.from_rotvec(a[9 + 6 * v:12 + 6 * v]).as_matrix()
        Rr = Rotation.from_rotvec(b[9 + 6 * v:12 + 6 * v]).as_matrix()
        Rs = Rr @ Rl.T
        rv.append(rotvec(Rs))
        tv.append(b[12 + 6 * v:15 + 6 * v] - Rs @ a[12 + 6 * v:15 + 6 * v])
    pose = np.r_[np.median(rv, axis=0), np.median(tv, axis=0)]
    return np.r_[a[:9], b[:9], a[9:], pose]
