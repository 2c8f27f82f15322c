"""CPU, world_size 2, gloo: the sharded iteration loop (slice bounds, the
32-double all-reduce, redundant solve on every rank) with a numpy engine in
place of the GPU one.  The engine lives here, in the tests, on purpose."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import refmath

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyShardEngine:
    def __init__(self, po, tgt, nrm, src_slice):
        self.po, self.tgt, self.nrm = po, tgt, nrm
        self.tree = po.KDTree(tgt)
        self.src0 = src_slice.copy()

    def begin(self):
        self.src = self.src0.copy()
        self.M = None
        self.trans = np.eye(4)
        self.iters = 0

    def accumulate(self):
        if self.M is not None:
            self.src = refmath.apply_update(self.src, self.M)
        if len(self.src):
            idx, d2 = self.tree.knn(self.src, 1)
            S = refmath.sums32(self.src, self.tgt, self.nrm, idx[:, 0], d2[:, 0])
        else:
            S = np.zeros(32)
        return torch.from_numpy(S)

    def update(self, summed):
        X = refmath.solve_from_sums(summed.numpy())
        self.M = refmath.euler_to_RT(X)
        self.trans = self.M @ self.trans
        self.iters += 1

    def finish(self):
        return self.trans, self.iters


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle import pyoracle as po
    from stereovision_amd.sharded import run_sharded, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "synth32.npz"))
    b, e = shard_bounds(len(g["source"]), world, rank)
    eng = NumpyShardEngine(po, g["target"], g["normals"], g["source"][b:e])
    T, iters = run_sharded(eng, 4)
    np.save(os.path.join(out_dir, f"T{rank}.npy"), T)
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from stereovision_amd.sharded import shard_bounds
    for n in (0, 1, 7, 1024, 16777216 + 3):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_ranks_match_the_unsharded_oracle(po, synth32, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    T0 = np.load(tmp_path / "T0.npy")
    T1 = np.load(tmp_path / "T1.npy")
    assert np.array_equal(T0, T1), "every rank must hold the same transform"
    r = po.icp_point2plane(po.KDTree(synth32["target"]), synth32["normals"], synth32["source"], 4)
    assert np.allclose(T0, r["trans"], rtol=0, atol=1e-9)


# ---- calibration: views split over the ranks ---------------------------------
SHARED = ("LL", "RR", "RS", "SS", "gL", "gR", "gS")


class OracleCalEngine:
    """CPU engine for evaluate_sharded: the oracle on this rank's views; the sums that
    cross ranks are the shared blocks and |e|^2, flattened."""

    def __init__(self, po, world, left, right):
        self.po, self.w, self.l, self.r, self.stereo = po, world, left, right, True

    def set_params(self, p, k):
        self.p, self.k = p, k

    def eval(self):
        self.out = self.po.stereo_eval_sparse(self.p, self.k, self.w, self.l, self.r, False)

    def export_sums(self):
        flat = [self.out[n].ravel() for n in SHARED] + [np.array([self.out["sq_residual"]])]
        return torch.from_numpy(np.concatenate(flat))

    def import_sums(self, t):
        v, o = t.numpy(), 0
        for n in SHARED:
            sz = self.out[n].size
            self.out[n] = v[o:o + sz].reshape(self.out[n].shape).copy()
            o += sz
        self.out["sq_residual"] = float(v[o])

    def fetch(self):
        return self.out

    def sq_residual(self):
        return self.out["sq_residual"]

    # the two halves of the damped step, as svcal_solve_begin / svcal_solve_end split it
    def solve_begin(self, lam):
        o, nI = self.out, 4 + self.k
        EE = o["EE"] + lam * np.einsum("vij,ij->vij", o["EE"], np.eye(6))
        rhs = np.concatenate([np.transpose(o["LE"], (0, 2, 1)), o["gE"][:, :, None]], axis=2)
        self.W = np.linalg.solve(EE, rhs)                          # (V, 6, nI + 1)
        contrib = np.einsum("vaq,vqc->ac", o["LE"], self.W)        # (nI, nI + 1)
        flat = np.zeros(90)
        flat[:nI * (nI + 1)] = contrib.ravel()
        return torch.from_numpy(flat)

    def solve_end(self, lam, t):
        o, nI = self.out, 4 + self.k
        csum = t.numpy()[:nI * (nI + 1)].reshape(nI, nI + 1)
        nC = 2 * nI + 6
        A, g = np.zeros((nC, nC)), np.zeros(nC)
        A[:nI, :nI], g[:nI] = o["LL"], o["gL"]
        A[nI:2 * nI, nI:2 * nI], A[nI:2 * nI, 2 * nI:] = o["RR"], o["RS"]
        A[2 * nI:, nI:2 * nI], A[2 * nI:, 2 * nI:] = o["RS"].T, o["SS"]
        g[nI:2 * nI], g[2 * nI:] = o["gR"], o["gS"]
        S = A + lam * np.diag(np.diag(A))
        S[:nI, :nI] -= csum[:, :nI]
        rhs = g.copy()
        rhs[:nI] -= csum[:, nI]
        dc = -np.linalg.solve(S, rhs)
        de = -(self.W[:, :, nI] + np.einsum("vqa,a->vq", self.W[:, :, :nI], dc[:nI]))
        step = np.concatenate([dc[:2 * nI], de.ravel(), dc[2 * nI:]])
        return step, float(dc @ (lam * dc - g)), float((de * (lam * de - o["gE"])).sum())


def _cal_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    from stereovision_amd import synth
    from stereovision_amd.calib import evaluate_sharded, view_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    V = 7
    w, l, r, _, start = synth.make_stereo_views(views=V)
    b, e = view_bounds(V, world, rank)
    eng = OracleCalEngine(po, w, l[b:e], r[b:e])
    out = evaluate_sharded(eng, start, 5, V, b, e)
    from stereovision_amd.calib import lm_refine_sharded, solve_damped_sharded
    step, pred = solve_damped_sharded(eng, 0.05, 5, V, b, e)
    x, err = lm_refine_sharded(eng, start, 5, V, b, e, max_iter=4)
    np.savez(os.path.join(out_dir, f"cal{rank}.npz"), begin=b, end=e,
             **{n: out[n] for n in SHARED + ("LE", "EE", "gE")}, sq=out["sq_residual"],
             step=step, pred=pred, x=x, err=err)
    dist.destroy_process_group()


def test_view_bounds_and_split():
    from stereovision_amd.calib import split_views, view_bounds
    for V in (1, 7, 2000):
        for w in (1, 2, 3, 8):
            spans = [view_bounds(V, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == V
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
    p = np.arange(18 + 6 * 5 + 6, dtype=np.float64)
    q = split_views(p, 5, 5, True, 1, 3)
    assert np.array_equal(q, np.r_[p[:18], p[24:36], p[-6:]])


def test_calibration_views_split_over_two_ranks(po, tmp_path):
    from stereovision_amd import synth
    port = _free_port()
    mp.spawn(_cal_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    w, l, r, _, start = synth.make_stereo_views(views=7)
    whole = po.stereo_eval_sparse(start, 5, w, l, r, False)
    parts = [np.load(tmp_path / f"cal{i}.npz") for i in range(2)]
    for n in SHARED:
        assert np.array_equal(parts[0][n], parts[1][n])
        assert np.allclose(parts[0][n], whole[n], rtol=1e-13, atol=1e-13 * np.abs(whole[n]).max())
    assert float(parts[0]["sq"]) == pytest.approx(whole["sq_residual"], rel=1e-13)
    for part in parts:
        b, e = int(part["begin"]), int(part["end"])
        for n in ("LE", "EE", "gE"):
            assert np.array_equal(part[n], whole[n][b:e])
    # the damped step and the refinement loop over the two ranks against the same on one
    from stereovision_amd.calib import lm_refine, solve_damped
    want, wpred = solve_damped(whole, 5, 7, 0.05)
    for part in parts:
        assert np.abs(part["step"] - want).max() < 1e-9 * np.abs(want).max()
        assert float(part["pred"]) == pytest.approx(wpred, rel=1e-9)
    assert np.array_equal(parts[0]["x"], parts[1]["x"])

    class Whole:
        V, stereo = 7, True

        def evaluate(self, p, k, want_residual=False):
            return po.stereo_eval_sparse(p, k, w, l, r, False)

    xw, ew = lm_refine(Whole(), start, 5, max_iter=4, host_solve=True)
    assert float(parts[0]["err"]) == pytest.approx(ew, rel=1e-6)
    assert np.abs(parts[0]["x"] - xw).max() < 1e-6 * np.abs(xw).max()


# ---- batches of independent pairs split over the ranks --------------------------
class OracleBatchEngine:
    """CPU engine for run_pairs_split: the oracle on each of this rank's pairs."""

    def __init__(self, po, pairs):
        self.po, self.pairs = po, pairs

    def run(self, iterations):
        T, info = [], []
        for tgt, nrm, src in self.pairs:
            r = self.po.icp_point2plane(self.po.KDTree(tgt), nrm, src, iterations)
            T.append(r["trans"])
            info.append([0.0, r["info"][-1, 0]])
        return np.array(T).reshape(-1, 4, 4), np.array(info).reshape(-1, 2)


def _make_pairs(count):
    from stereovision_amd import synth
    out = []
    for i in range(count):
        t, n, s, _ = synth.make_pair(24 + 4 * (i % 3), pair=i, pose=synth.small_pose(i))
        out.append((t, n, s))
    return out


def _batch_worker(rank, world, port, out_dir, count):
    sys.path.insert(0, ROOT)
    from oracle import pyoracle as po
    from stereovision_amd.batched import pair_bounds, run_pairs_split
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, last = pair_bounds(count, world, rank)
    eng = OracleBatchEngine(po, _make_pairs(count)[first:last])
    T, info = run_pairs_split(eng, count, 3)
    np.save(os.path.join(out_dir, f"BT{rank}.npy"), T)
    np.save(os.path.join(out_dir, f"BI{rank}.npy"), info)
    dist.destroy_process_group()


def test_pair_bounds_split_every_pair_once():
    from stereovision_amd.batched import pair_bounds
    for count in (0, 1, 5, 512, 4096):
        for w in (1, 2, 3, 8):
            spans = [pair_bounds(count, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(l - f for f, l in spans) - min(l - f for f, l in spans) <= 1
    assert pair_bounds(4096, 8, 3) == (1536, 2048)       # 512 pairs per GPU


def test_pairs_split_over_two_ranks_are_gathered_in_order(po, tmp_path):
    count = 5                                             # an odd count: ranks get 2 and 3 pairs
    port = _free_port()
    mp.spawn(_batch_worker, args=(2, port, str(tmp_path), count), nprocs=2, join=True)
    T0, T1 = np.load(tmp_path / "BT0.npy"), np.load(tmp_path / "BT1.npy")
    assert T0.shape == (count, 4, 4) and np.array_equal(T0, T1)
    assert np.array_equal(np.load(tmp_path / "BI0.npy"), np.load(tmp_path / "BI1.npy"))
    for i, (t, n, s) in enumerate(_make_pairs(count)):
        r = po.icp_point2plane(po.KDTree(t), n, s, 3)
        assert np.array_equal(T0[i], r["trans"])
