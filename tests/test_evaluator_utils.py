"""KITTI export (core/evaluator_utils.py) and the host-side angle-bin encoder against the oracle restatement."""
import os
import types

import numpy as np

from monopsr_amd.core import evaluator_utils, orientation_encoder
from oracle import geometry as G


def test_orientation_encoder_matches_oracle():
    rng = np.random.default_rng(0)
    for o in rng.uniform(-7, 7, 300):
        for overlap in (0.0, 0.2):
            a = orientation_encoder.np_orientation_to_angle_bin(o, 12, overlap)
            b = G.orientation_to_angle_bin(o, 12, overlap)
            assert a[0] == b[0]
            np.testing.assert_allclose(a[1], b[1], atol=1e-12)
            np.testing.assert_array_equal(a[2], b[2])
        k, r, _ = orientation_encoder.np_orientation_to_angle_bin(o, 12)
        assert abs(orientation_encoder.np_angle_bin_to_orientation(k, r[k], 12)
                   - G.angle_bin_to_orientation(k, r[k], 12)) < 1e-12
    np.testing.assert_allclose(orientation_encoder.np_wrap_to_pi(np.deg2rad([-185.0, 185.0, 90.0])),
                               np.deg2rad([175.0, -175.0, 90.0]), atol=1e-12)


def test_kitti_export(tmp_path):
    d3, d2 = tmp_path / "box_3d", tmp_path / "box_2d"
    d3.mkdir()
    d2.mkdir()
    b3 = np.array([[1.23456, 1.5, 20.0, 3.9, 1.6, 1.5, 0.31234, 0.9, 0], [2.0, 1.6, 30.0, 4.0, 1.7, 1.4, -1.0, 0.05, 0]])
    b2 = np.array([[150.0, 500.0, 220.0, 620.0, 0.25, 0.9, 0], [160.0, 700.0, 200.0, 760.0, -1.1, 0.05, 0]])
    np.savetxt(d3 / "000001.txt", b3, fmt='%0.5f')
    np.savetxt(d2 / "000001.txt", b2, fmt='%0.5f')
    np.savetxt(d3 / "000002.txt", b3[1:], fmt='%0.5f')      # only a below-threshold box
    np.savetxt(d2 / "000002.txt", b2[1:], fmt='%0.5f')
    ds = types.SimpleNamespace(data_split='val', num_samples=3, classes=['Car'],
                               sample_list=[types.SimpleNamespace(name=n) for n in ("000001", "000002", "000003")])
    out = evaluator_utils.save_predictions_box_3d_in_kitti_format(0.1, ds, str(tmp_path), str(d3), str(d2), 1234)
    assert out.endswith("kitti_predictions_3d/val/0.1/1234/data")
    lines = open(os.path.join(out, "000001.txt"), newline='').read().split('\r\n')
    assert lines[1] == ''
    f = lines[0].split()
    assert f[0] == 'Car' and f[1] == '-1' and f[2] == '-1' and len(f) == 16
    want = [0.25, 500.0, 150.0, 620.0, 220.0, 1.5, 1.6, 3.9, 1.235, 1.5, 20.0, 0.312, 0.9]
    np.testing.assert_allclose([float(v) for v in f[3:]], want, atol=1e-9)
    assert open(os.path.join(out, "000002.txt")).read() == ''   # filtered by score
    assert open(os.path.join(out, "000003.txt")).read() == ''   # no prediction file
