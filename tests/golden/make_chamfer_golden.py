"""Generates tests/golden/chamfer_sklearn.npz by importing the REFERENCE's Python Chamfer implementation.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_chamfer_golden.py

The reference function is monopsr.core.distance_metrics.calc_chamfer_dist
(/root/reference/src/monopsr/core/distance_metrics.py:5-23, sklearn KD-tree); the reference's own test
tf_nndistance_test.py:88-106 uses it as the independent check of sum(dist1)+sum(dist2).  The file written here
holds data only: seeded input clouds and the reference's float64 Chamfer sums for them.
"""
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference/src")
from monopsr.core import distance_metrics  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    rng = np.random.default_rng(20261001)
    cases = {}
    # (b, n, m, scale): ragged n != m, single points, the reference model's 48x48 map subsampled, KITTI-scale coords
    for name, (b, n, m, scale) in {
        "small": (3, 5, 7, 1.0),
        "single": (2, 1, 4, 3.0),
        "ragged": (2, 129, 64, 2.0),
        "cloud512": (2, 512, 512, 1.5),
        "wide": (1, 300, 1000, 40.0),
    }.items():
        xyz1 = (rng.standard_normal((b, n, 3)) * scale).astype(np.float32)
        xyz2 = (rng.standard_normal((b, m, 3)) * scale).astype(np.float32)
        sums = np.array([distance_metrics.calc_chamfer_dist(xyz1[i].astype(np.float64), xyz2[i].astype(np.float64))
                         for i in range(b)], dtype=np.float64)
        cases[name + "_xyz1"] = xyz1
        cases[name + "_xyz2"] = xyz2
        cases[name + "_chamfer"] = sums
    # the reference test's own vector (tf_nndistance_test.py:92-93) -> 164.0
    p1 = np.array([[[1., 1., 1.], [2., 2., 2.], [1., 5., 7.]]], np.float32)
    p2 = np.array([[[1., 5., 7.], [10., 0., 5.]]], np.float32)
    cases["reftest_xyz1"] = p1
    cases["reftest_xyz2"] = p2
    cases["reftest_chamfer"] = np.array([distance_metrics.calc_chamfer_dist(p1[0], p2[0])], np.float64)
    np.savez_compressed(os.path.join(HERE, "chamfer_sklearn.npz"), **cases)
    for k in sorted(cases):
        if k.endswith("_chamfer"):
            print(k, cases[k])


if __name__ == "__main__":
    main()
