"""CPU, world_size 2, gloo: the sharded iteration loop (slice bounds, the
32-double all-reduce, redundant solve on every rank) with a numpy engine in
place of the GPU one.  The engine lives here, in the tests, on purpose."""
import os
import socket
import sys

import numpy as np
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
