"""HIP geometry / map-loss kernels (geometry.hip) vs oracle/geometry.py and oracle/losses.py, and their gradients vs
float64 torch autograd of the same formulas.  SURVEY.md 8(f) rows 3-4."""
import math

import numpy as np
import pytest
import torch

from oracle import geometry as G
from oracle import losses as L

pytestmark = pytest.mark.gpu

P2 = np.array([[721.5377, 0, 609.5593, 44.85728], [0, 721.5377, 172.854, 0.2163791], [0, 0, 1, 0.002745884]],
              np.float32)


def _dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return (t if dtype is None else t.to(dtype)).cuda()


def _boxes(rng, n):
    y1, x1 = rng.uniform(0, 200, n), rng.uniform(0, 1000, n)
    return np.stack([y1, x1, y1 + rng.uniform(20, 170, n), x1 + rng.uniform(20, 220, n)], 1).astype(np.float32)


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


# ------------------------------------------------------------------------------------------ local -> global

def test_local_to_global_reference_vector():
    """instance_utils_test.py:52-73: 2304 random points, 10 degrees, centroid [2.5, 1.5, 15]."""
    from monopsr_amd.datasets.kitti import instance_utils as iu
    pts = np.random.default_rng(0).random((2304, 3)).astype(np.float32)
    va, cen = np.float32(np.deg2rad(10.0)), np.asarray([2.5, 1.5, 15.0], np.float32)
    want = G.inst_points_local_to_global(pts, va, cen)
    got = iu.tf_inst_xyz_map_local_to_global(_dev(pts.reshape(1, 48, 48, 3)), (48, 48), _dev(np.reshape(va, (1, 1))),
                                             _dev(cen.reshape(1, 3)))
    np.testing.assert_allclose(got.cpu().numpy().reshape(2304, 3), want, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("n,h,w", [(5, 48, 48), (3, 7, 9), (1, 1, 1), (0, 48, 48)])
def test_local_to_global_vs_oracle_and_grad(n, h, w):
    from monopsr_amd.datasets.kitti import instance_utils as iu
    rng = np.random.default_rng(n * 100 + h)
    x = rng.standard_normal((n, h, w, 3)).astype(np.float32) * 2
    va = rng.uniform(-0.8, 0.8, (n, 1)).astype(np.float32)
    cen = (rng.standard_normal((n, 3)) * 10).astype(np.float32)
    xt, ct = _dev(x).requires_grad_(), _dev(cen).requires_grad_()
    got = iu.tf_inst_xyz_map_local_to_global(xt, (h, w), _dev(va), ct)
    if n == 0:
        assert got.shape == (0, h, w, 3)
        return
    assert _rel(got.detach().cpu().numpy(), G.inst_xyz_map_local_to_global(x, va, cen)) < 2e-6
    g = rng.standard_normal((n, h, w, 3)).astype(np.float32)
    got.backward(_dev(g))
    x64 = torch.from_numpy(x).double().requires_grad_()
    c64 = torch.from_numpy(cen).double().requires_grad_()
    c, s = torch.cos(torch.from_numpy(va).double()).reshape(n, 1, 1), torch.sin(torch.from_numpy(va).double()).reshape(n, 1, 1)
    ref = torch.stack([c * x64[..., 0] + s * x64[..., 2], x64[..., 1], -s * x64[..., 0] + c * x64[..., 2]], -1) \
        + c64.reshape(n, 1, 1, 3)
    ref.backward(torch.from_numpy(g).double())
    assert _rel(xt.grad.cpu().numpy(), x64.grad.numpy()) < 2e-6
    assert _rel(ct.grad.cpu().numpy(), c64.grad.numpy()) < 1e-5


# --------------------------------------------------------------------------------------- projection error

def _proj_err_torch64(xyz, boxes, cam_p, mask):
    n, h, w, _ = xyz.shape
    b = torch.from_numpy(boxes).double()
    P = torch.from_numpy(cam_p).double()
    hu, hv = (b[:, 3] - b[:, 1]) / w / 2, (b[:, 2] - b[:, 0]) / h / 2
    ju, jv = torch.arange(w).double(), torch.arange(h).double()
    gu = (b[:, 1] + hu)[:, None] + ju[None] * ((b[:, 3] - b[:, 1] - 2 * hu) / max(w - 1, 1))[:, None]
    gv = (b[:, 0] + hv)[:, None] + jv[None] * ((b[:, 2] - b[:, 0] - 2 * hv) / max(h - 1, 1))[:, None]
    pad = torch.cat([xyz, torch.ones_like(xyz[..., :1])], -1)
    uvw = pad @ P.T
    u, v = uvw[..., 0] / uvw[..., 2], uvw[..., 1] / uvw[..., 2]
    m = torch.from_numpy(mask).double().reshape(n, h, w)
    eu = torch.clamp((gu[:, None, :] - u) / (b[:, 3] - b[:, 1])[:, None, None] * m, -2, 2)
    ev = torch.clamp((gv[:, :, None] - v) / (b[:, 2] - b[:, 0])[:, None, None] * m, -2, 2)
    nv = torch.clamp(m.sum((1, 2)), min=1.0)
    return (eu + ev).sum((1, 2)) / nv


@pytest.mark.parametrize("n,h,w", [(6, 48, 48), (2, 5, 7), (1, 1, 1)])
def test_proj_err_norm_vs_oracle_and_grad(n, h, w):
    from monopsr_amd.datasets.kitti import instance_utils as iu
    rng = np.random.default_rng(7 + n)
    boxes = _boxes(rng, n)
    xyz = np.empty((n, h, w, 3), np.float32)
    xyz[..., 2] = rng.uniform(6, 40, (n, h, w))
    xyz[..., 0] = rng.uniform(-0.5, 0.5, (n, h, w)) * xyz[..., 2]
    xyz[..., 1] = rng.uniform(-0.1, 0.15, (n, h, w)) * xyz[..., 2]
    xyz[0, 0, 0] = [500.0, 0.0, 5.0]            # far off: exercises the clip
    mask = (rng.uniform(size=(n, h, w, 1)) > 0.3).astype(np.float32)
    if n > 1:
        mask[1] = 0                              # an instance with no valid pixel divides by 1
    xt = _dev(xyz).requires_grad_()
    norm, maps = iu.proj_err_maps_norm(xt, _dev(boxes), _dev(P2), _dev(mask), want_maps=True)
    ref_norm, ref_maps = G.proj_err_maps_norm(xyz, boxes, P2, mask)
    assert np.abs(maps.cpu().numpy() - ref_maps).max() < 2e-5
    np.testing.assert_allclose(norm.detach().cpu().numpy(), ref_norm, rtol=2e-4, atol=2e-6)
    g = rng.standard_normal(n).astype(np.float32)
    norm.backward(_dev(g))
    x64 = torch.from_numpy(xyz).double().requires_grad_()
    r64 = _proj_err_torch64(x64, boxes, P2, mask)
    np.testing.assert_allclose(norm.detach().cpu().numpy(), r64.detach().numpy(), rtol=2e-4, atol=2e-6)
    r64.backward(torch.from_numpy(g).double())
    assert _rel(xt.grad.cpu().numpy(), x64.grad.numpy()) < 2e-5


# ------------------------------------------------------------------------------------- global depth maps

@pytest.mark.parametrize("rotate", [False, True])
@pytest.mark.parametrize("strided", [False, True])
def test_depth_map_global_vs_oracle_and_grad(rotate, strided):
    from monopsr_amd.datasets.kitti import instance_utils as iu
    rng = np.random.default_rng(11)
    n, h, w = 5, 48, 48
    boxes = _boxes(rng, n)
    xyz = rng.standard_normal((n, h, w, 3)).astype(np.float32)
    z = rng.uniform(6, 40, (n, 1)).astype(np.float32)
    va = rng.uniform(-0.6, 0.6, (n, 1)).astype(np.float32)
    xt = _dev(xyz).requires_grad_()
    zt = _dev(z).requires_grad_()
    d_in = xt[:, :, :, 2:3] if strided else xt[:, :, :, 2:3].contiguous()
    got = iu.tf_inst_depth_map_local_to_global(d_in, zt, _dev(boxes), _dev(va), (h, w), _dev(P2), rotate)
    ref = G.inst_depth_map_local_to_global(xyz[..., 2:3], z, boxes, va, P2, rotate)
    assert got.shape == (n, h, w, 1)
    assert np.abs(got.detach().cpu().numpy() - ref).max() < 3e-5
    g = rng.standard_normal((n, h, w, 1)).astype(np.float32)
    got.backward(_dev(g))
    gx = np.zeros_like(xyz)
    gx[..., 2:3] = g
    np.testing.assert_array_equal(xt.grad.cpu().numpy(), gx)
    ref64 = G.inst_depth_map_local_to_global(np.zeros((n, h, w, 1)), np.ones((n, 1)), boxes, va, P2, rotate, np.float64)
    want_gz = (g.astype(np.float64) * ref64).sum((1, 2, 3))      # the output is affine in z with slope ref64
    np.testing.assert_allclose(zt.grad.cpu().numpy().reshape(-1), want_gz, rtol=2e-4, atol=1e-4)


# --------------------------------------------------------------------------------- masked smooth-L1 sums

@pytest.mark.parametrize("shape", [(4, 48, 48, 3), (3, 9, 5, 1), (1, 32, 1)])
def test_nonzero_smooth_l1_vs_oracle_and_grad(shape):
    from monopsr_amd.core import losses_custom
    rng = np.random.default_rng(len(shape))
    p = (rng.standard_normal(shape) * 2).astype(np.float32)
    t = rng.standard_normal(shape).astype(np.float32)
    m = (rng.uniform(size=shape[:-1] + (1,)) > 0.3).astype(np.float32) * rng.choice([1.0, 0.5], shape[:-1] + (1,))
    m = m.astype(np.float32)
    pt = _dev(p).requires_grad_()
    loss = losses_custom.WeightedNonZeroSmoothL1LocalizationLoss()(pt, _dev(t), weights=_dev(m))
    np.testing.assert_allclose(float(loss), L.weighted_nonzero_smooth_l1(p, t, m), rtol=1e-5)
    (loss * 3.0).backward()
    p64 = torch.from_numpy(p).double().requires_grad_()
    e = (p64 - torch.from_numpy(t).double()).abs()
    q = torch.clamp(e, max=1.0)
    w64 = torch.from_numpy(m).double().expand(*shape)
    ref = ((0.5 * q * q + (e - q)) * w64).sum() / (w64 != 0).sum()
    (ref * 3.0).backward()
    assert _rel(pt.grad.cpu().numpy(), p64.grad.numpy()) < 1e-5
    zero = losses_custom.WeightedNonZeroSmoothL1LocalizationLoss()(_dev(p), _dev(t), weights=_dev(np.zeros_like(m)))
    assert float(zero) == 0.0


def test_loss_classes_vs_oracle_known_answers():
    """object_detection/core/losses_test.py:87-107 and :490-544 through the torch classes."""
    from monopsr_amd.core import losses
    pred = np.array([[[2.5, 0, .4, 0], [0, 0, 0, 0], [0, 2.5, 0, .4]], [[3.5, 0, 0, 0], [0, .4, 0, .9], [0, 0, 1.5, 0]]],
                    np.float32)
    w = np.array([[2, 1, 1], [0, 3, 0]], np.float32)
    got = losses.WeightedSmoothL1LocalizationLoss()(_dev(pred), _dev(np.zeros_like(pred)), weights=_dev(w))
    np.testing.assert_allclose(float(got.sum()), 7.695, rtol=1e-6)
    logits = np.array([[[-100, 100, -100], [100, -100, -100], [0, 0, -100], [-100, -100, 100]],
                       [[-100, 0, 0], [-100, 100, -100], [-100, 100, -100], [100, -100, -100]]], np.float32)
    tgt = np.array([[[0, 1, 0], [1, 0, 0], [1, 0, 0], [0, 0, 1]], [[0, 0, 1], [0, 1, 0], [0, 1, 0], [1, 0, 0]]],
                   np.float32)
    ww = np.array([[1, 1, .5, 1], [1, 1, 1, 0]], np.float32)
    got = losses.WeightedSoftmaxClassificationLoss()(_dev(logits), _dev(tgt), weights=_dev(ww)).cpu().numpy()
    np.testing.assert_allclose(got, [[0, 0, -0.5 * math.log(.5), 0], [-math.log(.5), 0, 0, 0]], atol=1e-6)


# ------------------------------------------------------------------------------------------ format boxes

def test_format_boxes_vs_oracle():
    from monopsr_amd.datasets.kitti import instance_utils as iu
    rng = np.random.default_rng(21)
    n, nb = 40, 12
    boxes = _boxes(rng, n)
    z = rng.uniform(4, 60, n)
    view = rng.uniform(-0.7, 0.7, n)
    cen = np.stack([z * np.tan(view) + rng.normal(0, 0.3, n), rng.uniform(0.5, 1.5, n), z], 1).astype(np.float32)
    cen[0] = [80.0, 1.0, 10.0]     # projects off the image: the 0.1 fit score branch
    cen[1] = [0.0, 1.0, 2.5]       # huge projection: discarded (> 80 % of the image)
    lwh = (np.array([3.9, 1.6, 1.5]) + rng.normal(0, 0.2, (n, 3))).astype(np.float32)
    bins = rng.standard_normal((n, nb)).astype(np.float32)
    regs = rng.uniform(-0.3, 0.3, (n, nb)).astype(np.float32)
    bins[2, 11], regs[2, 11] = 9.0, 0.26     # bin 11 + residual > pi: wraps
    scores = rng.uniform(0.1, 1.0, n).astype(np.float32)
    cls = np.ones((n, 1), np.int32)
    for centroid_type, post in (("middle", True), ("bottom", False)):
        b3, b2 = iu.format_boxes(_dev(lwh), _dev(view.astype(np.float32)), _dev(bins), _dev(regs), _dev(cen),
                                 _dev(boxes), _dev(scores), _dev(cls), _dev(P2), (375, 1242),
                                 centroid_type=centroid_type, post_process_cen_x=post)
        r3, r2 = G.format_predictions(lwh, view.astype(np.float32), bins, regs, cen, boxes, scores, cls, P2,
                                      (375, 1242), nb, centroid_type, post)
        np.testing.assert_allclose(b3.cpu().numpy(), r3, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(b2.cpu().numpy(), r2, rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------------- the model's loss

def test_model_train_outputs_and_loss_vs_oracle():
    """MonoPSRModel in 'train' mode on a 1/4-width net: global maps and every loss term vs the numpy restatement
    evaluated on the model's own head outputs; total = sum of terms; backward reaches the xyz-map head."""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    B, div = 5, 4
    cfg = config_utils.default_config()
    net = train_net.TrainNet(W.synthetic_weights(seed=71, width_div=div), width_div=div)
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config)
    rng = np.random.default_rng(72)
    boxes = _boxes(rng, B)
    sample = dict(rgb_image_crops=_dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)),
                  full_img_feature_crop=_dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0)
                                             .astype(np.float32)),
                  boxes_2d=_dev(boxes), cam_p=_dev(P2),
                  est_view_angs=_dev(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
                  class_indices=torch.ones((B, 1), dtype=torch.int32, device="cuda"),
                  mean_lwh=_dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                  prop_cen_z_offset=torch.full((B,), 2.178, device="cuda"))
    sample.update(trainer.synthetic_ground_truth(sample, seed=73))
    net.zero_grad()
    out = tr.forward(sample)
    losses_dict, total = tr.loss(out, sample)
    gtd = tr.model.gt_dict
    n = lambda t: t.detach().cpu().numpy()
    # global maps vs oracle on the model's own outputs
    gt_view = n(sample['gt_view_angs']).reshape(B, 1)
    cen = np.concatenate([n(out['cen_z']) * np.tan(gt_view) - P2[0, 3] / P2[0, 0], n(out['cen_y']), n(out['cen_z'])], 1)
    xyz_g = G.inst_xyz_map_local_to_global(n(out['inst_xyz_map_local']), gt_view, cen)
    pe, _ = G.proj_err_maps_norm(xyz_g, boxes, P2, n(sample['gt_valid_mask_maps']))
    np.testing.assert_allclose(n(out['proj_err_norm']), pe, rtol=1e-3, atol=1e-5)
    dg = G.inst_depth_map_local_to_global(n(out['inst_xyz_map_local'])[..., 2:3], n(out['cen_z']), boxes,
                                          n(out['view_ang']), P2, True)
    np.testing.assert_allclose(n(out['inst_depth_map_global']), dg, rtol=1e-5, atol=1e-4)
    # every loss term vs the restatement
    o = {k: n(v) for k, v in out.items() if torch.is_tensor(v)}
    g = {k: n(v) for k, v in gtd.items() if torch.is_tensor(v)}
    g['alpha_valid_bins'] = n(sample['gt_alpha_valid_bins'])
    lc = {k: v for k, v in cfg.model_config.loss_config.__dict__.items()}
    ref_d, ref_total = L.model_loss(o, g, B, lc)
    assert set(losses_dict) == set(ref_d)
    for k in ref_d:
        np.testing.assert_allclose(float(losses_dict[k]), ref_d[k], rtol=2e-4, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(float(total), ref_total, rtol=2e-4)
    total.backward()
    assert float(net.layers[net.n_trunk + net.n_dec - 1].dw.abs().max()) > 0   # xyz-map conv
    assert float(net.layers[0].dw.abs().max()) > 0                              # root conv


def test_trainer_schedule_and_moving_average():
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    cfg = config_utils.default_config()
    net = train_net.TrainNet(W.synthetic_weights(seed=81, width_div=8), width_div=8)
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config)
    lr = tr.optimizer.learning_rate
    assert lr(0) == 8e-5 and lr(9999) == 8e-5
    assert abs(lr(10000) - 8e-5 * 0.8) < 1e-12 and abs(lr(25000) - 8e-5 * 0.64) < 1e-12
    net.grads.normal_()
    p0 = net.params.clone()
    tr.optimizer.apply_gradients(net, 0)
    p1 = net.params.clone()
    assert torch.equal(tr.optimizer.shadow, p1) and not torch.equal(p0, p1)
    tr.optimizer.apply_gradients(net, 1)
    want = p1 * 0.9999 + net.params * 0.0001
    assert float((tr.optimizer.averaged_params(net) - want).abs().max()) < 1e-6


# ------------------------------------------------------------------------------------ argument validation

def test_geometry_entry_points_reject_bad_arguments():
    """Shape / pointer errors come back as MPSR_ERR_INVALID_ARG (raised as InvalidArgumentError), empty batches are
    no-ops -- the behaviour the reference gets from OP_REQUIRES in its op shells."""
    from monopsr_amd import _lib
    from monopsr_amd.core import losses_custom
    from monopsr_amd.datasets.kitti import instance_utils as iu
    lib = _lib.lib()
    x = torch.zeros((2, 4, 4, 3), device="cuda")
    s = _lib.stream()
    assert lib.mpsr_xyz_map_local_to_global(None, None, None, None, 0, 16, s) == 0          # b == 0: nothing to do
    assert lib.mpsr_xyz_map_local_to_global(None, None, None, None, 2, 16, s) == 1          # null pointers
    assert b"null pointer" in lib.mpsr_last_error()
    assert lib.mpsr_proj_err_norm(x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 2, 0, 4,
                                  s) == 1                                                   # h == 0
    assert lib.mpsr_depth_map_local_to_global(x.data_ptr(), 3, x.data_ptr(), None, None, None, x.data_ptr(), 2, 4, 4,
                                              1, s) == 1                                    # rotate_view needs boxes
    assert lib.mpsr_huber_loss_sums(x.data_ptr(), x.data_ptr(), x.data_ptr(), 2, 16, 3, 0.0, x.data_ptr(),
                                    x.data_ptr(), s) == 1                                   # delta must be > 0
    assert lib.mpsr_format_boxes(*([x.data_ptr()] * 9), 2, 0, 375, 1242, 1, 1, 45.0, x.data_ptr(), x.data_ptr(), s) == 1
    assert lib.mpsr_set_conv_math(7) == 1 and lib.mpsr_get_conv_math() == 0
    assert lib.mpsr_clip_by_norm_segments(x.data_ptr(), None, None, None, 3, x.data_ptr(), 5, 2, 1.0, s) == 1
    with pytest.raises(_lib.InvalidArgumentError):
        iu.tf_inst_xyz_map_local_to_global(x[..., :2], (4, 4), torch.zeros((2, 1), device="cuda"),
                                           torch.zeros((2, 3), device="cuda"))
    with pytest.raises(_lib.InvalidArgumentError):
        iu.tf_inst_depth_map_local_to_global(x[..., :1], torch.zeros((2, 1), device="cuda"), rotate_view=True)
    with pytest.raises(_lib.InvalidArgumentError):
        losses_custom.WeightedNonZeroSmoothL1LocalizationLoss()(x, x, weights=torch.ones((2, 4, 4), device="cuda"))
    with pytest.raises(_lib.MpsrError):
        iu.tf_inst_xyz_map_local_to_global(x.cpu(), (4, 4), torch.zeros((2, 1)), torch.zeros((2, 3)))   # no CPU path
