"""BASELINE config 1 on real data: 32 KITTI proposal crops (48x48) + 512-point GT clouds, extracted from the
reference's own mini-KITTI test fixture by tests/golden/make_kitti_fixture.py (committed as kitti_cfg1.npz).

CPU part: the oracle's Chamfer equals the REFERENCE's Python Chamfer (calc_chamfer_dist, values stored in the
fixture) on the real clouds.  GPU part: the HIP path vs the oracle on the real crops / clouds.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ops as orc


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "kitti_cfg1.npz"))


def test_fixture_shapes(fx):
    assert fx["rgb_crops"].shape == (32, 48, 48, 3) and fx["gt_clouds"].shape == (32, 512, 3)
    assert fx["boxes_2d"].shape == (32, 4) and fx["cam_p"].shape == (32, 3, 4)
    assert fx["full_frame"].dtype == np.uint8 and fx["full_frame"].shape[2] == 3
    assert np.isfinite(fx["rgb_crops"]).all() and np.abs(fx["rgb_crops"]).max() < 256


def test_oracle_chamfer_equals_reference_python_on_kitti_clouds(fx):
    d1, _, d2, _ = orc.nn_distance(fx["pred_clouds"], fx["gt_clouds"])
    got = d1.astype(np.float64).sum(1) + d2.astype(np.float64).sum(1)
    np.testing.assert_allclose(got, fx["chamfer_ref"], rtol=5e-6)


def test_oracle_emd_on_kitti_clouds_is_a_transport_plan(fx):
    p, g = fx["pred_clouds"][:4], fx["gt_clouds"][:4]
    for sem, axis in (("cpu", 2), ("gpu", 1)):
        m = orc.approx_match(p, g, sem)
        assert (m >= 0).all()
        np.testing.assert_allclose(m.sum(axis), 1.0, atol=3e-2)
    cc = orc.match_cost(p, g, orc.approx_match(p, g, "cpu"), "cpu")
    cg = orc.match_cost(p, g, orc.approx_match(p, g, "gpu"), "gpu")
    np.testing.assert_allclose(cg, cc, rtol=0.05)


# ------------------------------------------------------------------------------------------------ GPU

def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_cfg1_ops_on_gpu(fx):
    from monopsr_amd.tf_ops.approxmatch import tf_approxmatch as am
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance
    p, g = fx["pred_clouds"], fx["gt_clouds"]
    got = [t.cpu().numpy() for t in tf_nndistance.nn_distance(_dev(p), _dev(g))]
    ref = orc.nn_distance(p, g)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)  # dist and idx bit-exact
    np.testing.assert_allclose(got[0].astype(np.float64).sum(1) + got[2].astype(np.float64).sum(1),
                               fx["chamfer_ref"], rtol=5e-6)  # and equal to the reference's Python
    match = am.approx_match(_dev(p), _dev(g))
    rm = orc.approx_match(p, g, "gpu")
    np.testing.assert_allclose(match.cpu().numpy(), rm, rtol=1e-3, atol=1e-3 * rm.max())
    np.testing.assert_allclose(am.match_cost(_dev(p), _dev(g), match).cpu().numpy(),
                               orc.match_cost(p, g, rm, "gpu"), rtol=1e-3)


@pytest.mark.gpu
def test_cfg1_network_on_gpu(fx):
    """32 real crops through the full-width trunk / decoder / heads vs the CPU restatement (seeded weights)."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    from oracle import net as onet
    B = 32
    weights = W.synthetic_weights(seed=0)
    rng = np.random.default_rng(1)
    full_feat = np.maximum(rng.standard_normal((B, 12, 12, 1024)), 0).astype(np.float32)
    cls = np.ones((B, 1), np.int32)
    mean_lwh = np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))
    z_off = np.full((B,), 2.17799973487854, np.float32)
    cam_p = fx["cam_p"][0]
    ref = onet.instance_path(fx["rgb_crops"], full_feat, fx["boxes_2d"], cam_p, fx["view_angs"], cls, mean_lwh, z_off,
                             weights)
    net = dn.DeviceNet(weights)
    feat = net.trunk(_dev(fx["rgb_crops"]))
    fb, fm, xyz = net.squash_decoder(feat, _dev(full_feat))
    out = net.heads_fwd(fb, _dev(fx["boxes_2d"]), _dev(cam_p), _dev(fx["view_angs"]), _dev(cls), _dev(mean_lwh),
                        _dev(z_off))
    for got, want, name in ((feat, ref["crop_feat"], "block3"), (xyz, ref["inst_xyz_map_local"], "xyz"),
                            (out["centroids"], ref["centroids"], "centroids"), (out["lwh"], ref["lwh"], "lwh"),
                            (out["alpha_bins"], ref["alpha_bins"], "alpha_bins")):
        want = want.numpy()
        err = np.abs(got.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30)
        assert err < 1e-4, "%s drift %.3e" % (name, err)


@pytest.mark.gpu
def test_cfg1_image_to_crops_on_real_pixels(fx):
    """Preprocess + crop_and_resize on a real KITTI frame: the HIP crops equal the fixture's (oracle-made) crops."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core.img_preprocessor import ImgPreprocessor
    k = int(fx["full_frame_index"])
    rows = np.nonzero(fx["frame_index"] == k)[0]
    img = fx["full_frame"]
    H, Wd = img.shape[:2]
    pre = ImgPreprocessor().preprocess_input(_dev(img.astype(np.float32)).unsqueeze(0), (320, 1216), "kitti")
    norm = fx["boxes_2d"][rows] / np.array([H, Wd, H, Wd], np.float32)
    crops = dn.crop_and_resize(pre, _dev(norm), None, (48, 48)).cpu().numpy()
    np.testing.assert_allclose(crops, fx["rgb_crops"][rows], rtol=0, atol=2e-4)  # pixel scale 0..255
