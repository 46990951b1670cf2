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


def test_kitti_export_from_memory_matches_the_file_route(tmp_path):
    rng = np.random.default_rng(3)
    preds = {}
    d3, d2 = tmp_path / "b3", tmp_path / "b2"
    d3.mkdir()
    d2.mkdir()
    for i in range(4):
        n = int(rng.integers(0, 6))
        b3 = np.round(np.column_stack([rng.uniform(-20, 20, n), rng.uniform(1, 2, n), rng.uniform(5, 60, n),
                                       rng.uniform(3, 5, n), rng.uniform(1.4, 2, n), rng.uniform(1.3, 1.8, n),
                                       rng.uniform(-3.1, 3.1, n), rng.uniform(0, 1, n), np.zeros(n)]), 5)
        b2 = np.round(np.column_stack([rng.uniform(0, 300, n), rng.uniform(0, 1200, n), rng.uniform(0, 370, n),
                                       rng.uniform(0, 1240, n), rng.uniform(-3, 3, n), b3[:, 7], np.zeros(n)]), 5)
        name = "%06d" % i
        if n:
            preds[name] = (b3, b2)
            np.savetxt(d3 / (name + ".txt"), b3, fmt='%0.5f')
            np.savetxt(d2 / (name + ".txt"), b2, fmt='%0.5f')
    names = ["%06d" % i for i in range(4)]
    ds = types.SimpleNamespace(data_split='val', num_samples=4, classes=['Car'],
                               sample_list=[types.SimpleNamespace(name=n) for n in names])
    out_files = evaluator_utils.save_predictions_box_3d_in_kitti_format(0.3, ds, str(tmp_path), str(d3), str(d2), 7)
    out_mem = str(tmp_path / "mem")
    evaluator_utils.export_kitti_labels(preds, ['Car'], 0.3, out_mem, names)
    for n in names:
        assert open(os.path.join(out_files, n + ".txt"), newline='').read() == \
            open(os.path.join(out_mem, n + ".txt"), newline='').read()


def _project_one(box, cam_p, size):
    """Independent per-box loop version of box_3d_projector.py:14-95 (truncate=True, defaults otherwise)."""
    x, y, z, l, w, h, ry = box[:7]
    pts = []
    for sx, sy, sz in ((1, 0, 1), (1, 0, -1), (-1, 0, -1), (-1, 0, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1),
                       (-1, -1, 1)):
        cx, cy, cz = sx * l / 2, sy * h, sz * w / 2
        X = np.cos(ry) * cx + np.sin(ry) * cz + x
        Z = -np.sin(ry) * cx + np.cos(ry) * cz + z
        p = cam_p @ np.array([X, cy + y, Z, 1.0])
        pts.append(p[:2] / p[2])
    pts = np.array(pts)
    x1, y1, x2, y2 = pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()
    if x1 > size[0] or y1 > size[1] or x2 < 0 or y2 < 0:
        return None
    if x2 - x1 > 0.8 * size[0] or y2 - y1 > 0.8 * size[1]:
        return None
    return [max(x1, 0), max(y1, 0), min(x2, size[0]), min(y2, size[1])]


def test_projected_boxes_match_a_per_box_loop(tmp_path):
    cam_p = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
                      [0.0, 0.0, 1.0, 0.002745884]])
    size = (1242, 375)
    rng = np.random.default_rng(11)
    n = 200
    b3 = np.column_stack([rng.uniform(-30, 30, n), rng.uniform(1, 2.5, n), rng.uniform(2.5, 70, n),
                          rng.uniform(3, 5, n), rng.uniform(1.4, 2, n), rng.uniform(1.3, 1.8, n),
                          rng.uniform(-3.1, 3.1, n), rng.uniform(0, 1, n), np.zeros(n)])
    boxes, keep = evaluator_utils.project_boxes_3d(b3, cam_p, size)
    ref = [_project_one(b, cam_p, size) for b in b3]
    assert [r is not None for r in ref] == keep.tolist()
    assert 20 < keep.sum() < n  # the sample exercises both outcomes
    for r, got, k in zip(ref, boxes, keep):
        if k:
            np.testing.assert_allclose(got, r, rtol=0, atol=1e-9)
    # through the exporter: the projected box replaces columns 4..7, dropped boxes vanish
    b2 = np.column_stack([np.zeros((n, 4)), rng.uniform(-3, 3, n), b3[:, 7], np.zeros(n)])
    cnt = evaluator_utils.export_kitti_labels({"f": (b3, b2)}, ['Car'], 0.0, str(tmp_path), project_3d_box=True,
                                              frame_info=lambda name: (cam_p, size))
    assert cnt == 1
    lines = [l for l in open(tmp_path / "f.txt", newline='').read().split('\r\n') if l]
    assert len(lines) == int(keep.sum())
    first = int(np.flatnonzero(keep)[0])
    np.testing.assert_allclose([float(v) for v in lines[0].split()[4:8]], np.round(boxes[first], 3), atol=1e-9)
