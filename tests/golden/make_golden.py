"""Regenerates the golden fixtures under tests/golden/.

Run in the authoring container (needs /root/reference for the two bundled
clouds and the built CPU oracle):  python tests/golden/make_golden.py

cloud15.npy / cloud17.npy are the reference's own data files
(PointCloud/Registration/15.txt, 17.txt) parsed as CTxtReader does
(atof -> float) and stored as float32.  Everything else is produced by the CPU
oracle in oracle/icp_oracle.c.  The reference records no expected outputs of
its own, so these vectors pin the oracle against drift, not against the
reference ("parity unpinned").
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from stereovision_amd import synth  # noqa: E402

REF = "/root/reference/PointCloud/Registration"


def main():
    if os.path.isdir(REF):
        for name in ("15", "17"):
            raw = open(os.path.join(REF, name + ".txt"), "rb").read()
            cloud = po.parse_txt(raw)
            alt = np.loadtxt(os.path.join(REF, name + ".txt")).astype(np.float32)
            assert np.array_equal(cloud, alt)
            np.save(os.path.join(HERE, f"cloud{name}.npy"), cloud)
    ref = np.load(os.path.join(HERE, "cloud17.npy"))
    src = np.load(os.path.join(HERE, "cloud15.npy"))

    tree = po.KDTree(ref)
    normals, knn20 = po.calc_normals(tree, want_knn=True)
    r5 = po.icp_point2plane(tree, normals, src, 5)      # test.cpp:23 runs 5 iterations
    r30 = po.icp_point2plane(tree, normals, src, 30, history=False)
    nn_idx, nn_d2 = tree.knn(src, 1)
    np.savez_compressed(
        os.path.join(HERE, "pair_15_17.npz"),
        normals=normals,
        knn20_rowsum=knn20.astype(np.int64).sum(axis=1),
        knn20_head=knn20[:64],
        nn_idx=nn_idx[:, 0], nn_d2=nn_d2[:, 0],
        trans5=r5["trans"], X5=r5["X"], info5=r5["info"], idx5=r5["idx"], src5=r5["src"],
        trans30=r30["trans"], X30=r30["X"], info30=r30["info"],
    )

    # small synthetic pair with a known pose
    tgt, nrm, s, P = synth.make_pair(32)
    t2 = po.KDTree(tgt)
    r = po.icp_point2plane(t2, nrm, s, 10)
    np.savez_compressed(
        os.path.join(HERE, "synth32.npz"),
        target=tgt, normals=nrm, source=s, pose=P,
        trans10=r["trans"], X10=r["X"], idx0=r["idx"][0],
    )

    # tie case: unit lattice, queries at cell centres and on points
    g = np.arange(6, dtype=np.float32)
    lat = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    q = np.concatenate([lat[:50] + 0.5, lat[100:150], lat[20:40] + np.float32([0.5, 0, 0])]).astype(np.float32)
    bi, bd = po.knn_brute(lat, q, 8)
    np.savez_compressed(os.path.join(HERE, "lattice_ties.npz"), target=lat, query=q, idx=bi, d2=bd)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
