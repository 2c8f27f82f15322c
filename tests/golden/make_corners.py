"""Writes tests/golden/testdata_corners.npz from the corner tables of the reference's
Test/testdata.h (numbers only) and the board geometry of Test/demo.cpp:17-27 (8 x 6 board
for the single-camera case) and Test/demo.cpp:76-139 (9 x 6 board for the stereo case), both
with 25 mm squares, x running fastest.  Run in the build container; the .npz is committed.
"""
import os
import re
import sys

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/Test/testdata.h"
txt = open(REF).read()
arrs = {}
for m in re.finditer(r"Data_Type\s+(\w+)\[\]\[2\]\s*=\s*\{(.*?)\};", txt, re.S):
    nums = re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", m.group(2))
    arrs[m.group(1)] = np.array([float(x) for x in nums]).reshape(-1, 2)
left = np.stack([arrs[f"left_data{i}"] for i in range(1, 5)])
right = np.stack([arrs[f"right_data{i}"] for i in range(1, 5)])
mono = np.stack([arrs[f"data{i}"] for i in range(4)])


def board(nx, ny):
    gx, gy = np.meshgrid(np.arange(nx) * 25.0, np.arange(ny) * 25.0)
    return np.stack([gx.ravel(), gy.ravel()], axis=1)


out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "testdata_corners.npz")
np.savez_compressed(out, left=left, right=right, mono=mono, world_stereo=board(9, 6),
                    world_mono=board(8, 6))
print(out, left.shape, right.shape, mono.shape)
