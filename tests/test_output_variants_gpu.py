"""Output-type variants of the output builder beyond monopsr_model_000.yaml's set
(monopsr_output_builder.py:110-120 valid-mask head, :276-393 alpha 'dc_rotation' / 'prob' / 'gt', :441-507 cen_z
'direct', :509-549 view_ang 'offset' / 'gt', :573-609 cen_y 'gt' / 'est', :625-661 lwh 'gt', :774-803 xyz from depth)
and the loss terms that go with them (monopsr_model.py:597-620, :712-757).  The layers run through the HIP kernels;
the checker is a float64 restatement of each formula written out here."""
import math
import types

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

B, FEAT = 5, 64


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, ref, tol=1e-4, name=""):
    got = got.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(got) else np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = np.abs(got - ref).max() / max(1e-6, np.abs(ref).max())
    assert err <= tol, "%s: rel err %.3e" % (name, err)


@pytest.fixture(scope="module")
def setup():
    from monopsr_amd.core import constants
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(5)
    weights = W.synthetic_weights(seed=31, width_div=2)

    def fc(name, fin, fout):
        weights[name + "/weights"] = (rng.standard_normal((fin, fout)) * 0.2).astype(np.float32)
        weights[name + "/biases"] = rng.standard_normal(fout).astype(np.float32)
    fc("output/view_ang/view_ang", FEAT, 1)
    fc("output/cen_z_direct/cen_z", FEAT, 1)
    weights["output/valid_mask_maps/valid_mask_maps/weights"] = (rng.standard_normal((3, 3, 64, 1)) * 0.1).astype(np.float32)
    weights["output/valid_mask_maps/valid_mask_maps/biases"] = rng.standard_normal(1).astype(np.float32)
    net = dn.DeviceNet(weights, width_div=2)
    feats = {constants.FEATURES_FOR_MAP: _dev(rng.standard_normal((B, 48, 48, 64)).astype(np.float32)),
             constants.FEATURES_FOR_BOX_3D: _dev(rng.standard_normal((B, 6, 6, 256)).astype(np.float32))}
    cam_p = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
                      [0.0, 0.0, 1.0, 0.002745884]], np.float32)
    return types.SimpleNamespace(weights=weights, net=net, feats=feats, cam_p=cam_p, rng=rng, fc=fc,
                                 x=rng.standard_normal((B, FEAT)).astype(np.float32))


def _builder(setup, mode='train', **output_config):
    from monopsr_amd.core.models.monopsr.monopsr_output_builder import MonoPSROutputBuilder
    setup.net._fc_cache.clear()
    dataset_config = types.SimpleNamespace(num_alpha_bins=12, classes=['Car'])
    return MonoPSROutputBuilder(output_config, types.SimpleNamespace(), dataset_config, dict(setup.feats), B, (48, 48),
                                _dev(setup.cam_p), train_val_test=mode, device_net=setup.net)


def _fc_ref(setup, name, x):
    return x.astype(np.float64) @ setup.weights[name + "/weights"].astype(np.float64) + \
        setup.weights[name + "/biases"].astype(np.float64)


def test_valid_mask_head(setup):
    from monopsr_amd.core import constants
    b = _builder(setup)
    b.add_valid_mask_maps_output()
    x = setup.feats[constants.FEATURES_FOR_MAP].cpu().double().permute(0, 3, 1, 2)
    w = torch.from_numpy(setup.weights["output/valid_mask_maps/valid_mask_maps/weights"]).double().permute(3, 2, 0, 1)
    ref = torch.nn.functional.conv2d(x, w, torch.from_numpy(
        setup.weights["output/valid_mask_maps/valid_mask_maps/biases"]).double(), padding=1).permute(0, 2, 3, 1)
    _close(b.get_output_dict()[constants.KEY_VALID_MASK_MAPS], ref.numpy(), 1e-5, "valid mask logits")


def test_alpha_variants(setup):
    from monopsr_amd.core import constants
    gt_bins = _dev(setup.rng.integers(0, 12, (B, 1)).astype(np.float32))
    gt_regs = _dev(setup.rng.standard_normal((B, 12)).astype(np.float32))
    gt_alpha = _dev(setup.rng.uniform(-3, 3, (B, 1)).astype(np.float32))
    # dc_rotation: 12 bins + 12 (cos, sin) pairs
    setup.fc("output/alpha", FEAT, 36)
    b = _builder(setup, alpha='dc_rotation')
    b.add_alpha_output(_dev(setup.x), gt_alpha, [gt_bins, gt_regs])
    o = _fc_ref(setup, "output/alpha", setup.x)
    comp = o[:, 12:].reshape(B, 12, 2)
    comp = comp / np.sqrt(np.maximum((comp ** 2).sum(2, keepdims=True), 1e-12))
    _close(b.get_output_dict()[constants.KEY_ALPHA_BINS], o[:, :12], 1e-5, "dc_rotation bins")
    _close(b.get_output_dict()[constants.KEY_ALPHA_REGS], np.arctan2(comp[:, :, 1], comp[:, :, 0]), 1e-5, "dc_rotation regs")
    assert b.get_gt_dict()[constants.KEY_ALPHA_REGS] is gt_regs
    # prob
    setup.fc("output/alpha", FEAT, 12)
    b = _builder(setup, alpha='prob')
    b.add_alpha_output(_dev(setup.x), gt_alpha, [gt_bins, gt_regs])
    o = _fc_ref(setup, "output/alpha", setup.x)
    sm = np.exp(o - o.max(1, keepdims=True))
    sm /= sm.sum(1, keepdims=True)
    half = 2 * math.pi / 12 / 2
    centres = np.linspace(half, 2 * math.pi - half, 12)
    c = sm @ np.stack((np.cos(centres), np.sin(centres)), 1)
    _close(b.get_output_dict()[constants.KEY_ALPHA], np.arctan2(c[:, 1], c[:, 0])[:, None], 1e-5, "prob alpha")
    assert b.get_gt_dict()[constants.KEY_ALPHA] is gt_alpha
    # gt
    b = _builder(setup, alpha='gt')
    b.add_alpha_output(_dev(setup.x), gt_alpha, [gt_bins, gt_regs])
    assert b.get_output_dict()[constants.KEY_ALPHA_BINS] is gt_bins
    with pytest.raises(ValueError):
        _builder(setup, alpha='nonsense').add_alpha_output(_dev(setup.x), gt_alpha, [gt_bins, gt_regs])


def test_scalar_head_variants(setup):
    x = _dev(setup.x)
    est = _dev(setup.rng.uniform(-0.5, 0.5, (B, 1)).astype(np.float32))
    gt = _dev(setup.rng.uniform(-0.5, 0.5, (B, 1)).astype(np.float32))
    # view_ang 'offset' and 'gt'
    b = _builder(setup, view_ang='offset')
    b.add_view_ang_output('view_ang', x, est, gt)
    off = _fc_ref(setup, "output/view_ang/view_ang", setup.x)
    _close(b.get_output_dict()['view_ang_offs'], off, 1e-5, "view_ang offsets")
    _close(b.get_output_dict()['view_ang'], est.cpu().numpy() + off, 1e-5, "view_ang")
    _close(b.get_gt_dict()['view_ang_offs'], (gt - est).cpu().numpy(), 1e-6)
    b = _builder(setup, view_ang='gt')
    b.add_view_ang_output('view_ang', x, est, gt)
    assert b.get_output_dict()['view_ang'] is gt
    # cen_z 'direct': the FC output is the depth, no offset entry
    prop = _dev(setup.rng.uniform(5, 40, (B, 1)).astype(np.float32))
    b = _builder(setup, cen_z='direct')
    b.add_cen_z_output('cen_z', x, prop, gt)
    _close(b.get_output_dict()['cen_z'], _fc_ref(setup, "output/cen_z_direct/cen_z", setup.x), 1e-5, "cen_z direct")
    assert 'cen_z_offs' not in b.get_output()
    # cen_y 'gt' / 'est' (the reference's 'est' branch fails with a NameError)
    b = _builder(setup, cen_y='gt')
    b.add_cen_y_output('cen_y', x, prop, gt)
    assert b.get_output_dict()['cen_y'] is gt
    _close(b.get_output_dict()['cen_y_offs'], (gt - prop).cpu().numpy(), 1e-6)
    with pytest.raises(NameError):
        _builder(setup, cen_y='est').add_cen_y_output('cen_y', x, prop, gt)
    # lwh 'gt'
    est_lwh = _dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1)))
    gt_lwh = _dev(setup.rng.uniform(1, 4, (B, 3)).astype(np.float32))
    b = _builder(setup, lwh='gt')
    assert b.add_lwh_output(x, est_lwh, gt_lwh) is gt_lwh
    _close(b.get_output_dict()['lwh_offs'], (gt_lwh - est_lwh).cpu().numpy(), 1e-6)
    _close(b.get_gt_dict()['lwh_offs'], np.zeros((B, 3)), 1e-6)


def test_xyz_map_global_from_depth(setup):
    """:774-803 with tf_depth_patch_to_pc_map (depth_map_utils.py:161-236): pixel-centre grid of each box,
    x = (u - cu) d / f, y = (v - cv) d / f, z = d, then the reference's reshape of (3, h, w) to (h, w, 3)."""
    from monopsr_amd.core import constants
    rng = setup.rng
    depth = rng.uniform(5, 40, (B, 48, 48, 1)).astype(np.float32)
    y1, x1 = rng.uniform(0, 150, B), rng.uniform(0, 1000, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(20, 200, B), x1 + rng.uniform(20, 200, B)], 1).astype(np.float32)
    gt = _dev(rng.standard_normal((B, 48, 48, 3)).astype(np.float32))
    b = _builder(setup, inst_xyz_map_global_from_depth='x')
    b.add_inst_xyz_maps_global_from_depth(_dev(depth), _dev(boxes), gt)
    f, cu, cv = [float(v) for v in (setup.cam_p[0, 0], setup.cam_p[0, 2], setup.cam_p[1, 2])]
    ref = np.zeros((B, 48, 48, 3))
    for i in range(B):
        by1, bx1, by2, bx2 = [float(v) for v in boxes[i]]
        hw, hh = (bx2 - bx1) / 48 / 2, (by2 - by1) / 48 / 2
        xx, yy = np.meshgrid(np.linspace(bx1 + hw, bx2 - hw, 48), np.linspace(by1 + hh, by2 - hh, 48))
        d = depth[i, :, :, 0].astype(np.float64)
        pc = np.stack(((xx - cu) * d / f, (yy - cv) * d / f, d), 0)
        ref[i] = pc.reshape(48, 48, 3)
    _close(b.get_output_dict()[constants.KEY_INST_XYZ_MAP_GLOBAL_FROM_DEPTH], ref, 2e-5, "xyz from depth")
    assert b.get_gt_dict()[constants.KEY_INST_XYZ_MAP_GLOBAL_FROM_DEPTH] is gt


def test_valid_mask_and_prob_alpha_loss_terms(setup):
    """monopsr_model.py:597-620 (sigmoid cross entropy against label-smoothed masks, mean over pixels, sum over
    instances) and :712-752 (alpha 'prob': temperature softmax on hard one-hot bins + smooth-L1 on alpha)."""
    from monopsr_amd.core import config_utils, constants
    from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
    from oracle import losses as L
    cfg = config_utils.default_config()
    model = MonoPSRModel(cfg.model_config, cfg.dataset_config, setup.net, 'train', fused_heads=False)
    model.output_types = [constants.KEY_VALID_MASK_MAPS, constants.KEY_ALPHA]
    model.output_config.alpha = 'prob'
    model.num_boxes = B
    lc = model.model_config.loss_config
    setattr(lc, constants.KEY_VALID_MASK_MAPS, ['sigmoid_ce', 2.0])
    setattr(lc, constants.KEY_ALPHA + '_cls_temp', ['softmax_temp', 1.5])
    setattr(lc, constants.KEY_ALPHA + '_reg', ['smooth_l1', 0.7])
    rng = setup.rng
    logits = rng.standard_normal((B, 48, 48, 1)).astype(np.float32) * 3
    mask = (rng.random((B, 48, 48, 1)) > 0.5).astype(np.float32)
    bins = rng.standard_normal((B, 12)).astype(np.float32)
    gt_bins = rng.integers(0, 12, (B, 1))
    alpha, gt_alpha = rng.uniform(-3, 3, (B, 1)).astype(np.float32), rng.uniform(-3, 3, (B, 1)).astype(np.float32)
    out = {constants.KEY_VALID_MASK_MAPS: _dev(logits), constants.KEY_ALPHA_BINS: _dev(bins),
           constants.KEY_ALPHA: _dev(alpha), constants.KEY_LWH: _dev(np.zeros((B, 3), np.float32))}
    gt = {constants.KEY_VALID_MASK_MAPS: _dev(mask), constants.KEY_ALPHA_BINS: _dev(gt_bins.astype(np.float32)),
          constants.KEY_ALPHA: _dev(gt_alpha)}
    losses_dict, total = model.loss(out, gt)
    ref_mask = 2.0 * (L.sigmoid_ce(logits, mask * 0.998 + 0.001).sum(axis=(1, 2)) / (48 * 48)).sum()
    one_hot = np.eye(12)[gt_bins[:, 0]]
    ref_bins = 1.5 * L.weighted_softmax(bins[None], one_hot[None], np.ones((1, B)), logit_scale=0.5).sum() / B
    ref_reg = 0.7 * L.weighted_smooth_l1(alpha[None], gt_alpha[None], np.ones((1, B, 1))).sum() / B
    _close(losses_dict[constants.KEY_VALID_MASK_MAPS], ref_mask, 2e-5, "valid mask loss")
    _close(losses_dict[constants.KEY_ALPHA_BINS], ref_bins, 2e-5, "alpha bins (temperature softmax)")
    _close(losses_dict[constants.KEY_ALPHA], ref_reg, 2e-5, "alpha regression")
    _close(total, ref_mask + ref_bins + ref_reg, 2e-5, "total")


def test_variants_outside_model_000_are_refused_by_the_trainable_net(setup):
    """on a TrainNet the variant heads must fail loudly -- their layers are not in the flat parameter
    buffer, so running them would add loss terms that carry no gradient."""
    from monopsr_amd import _lib
    from monopsr_amd.core import train_net
    from monopsr_amd.core.models.monopsr.monopsr_output_builder import MonoPSROutputBuilder
    tnet = train_net.TrainNet(setup.weights, width_div=2)
    dataset_config = types.SimpleNamespace(num_alpha_bins=12, classes=['Car'])
    ob = MonoPSROutputBuilder({'valid_mask_maps': 1}, types.SimpleNamespace(), dataset_config, dict(setup.feats), B,
                              (48, 48), _dev(setup.cam_p), train_val_test='train', device_net=tnet)
    with pytest.raises(_lib.InvalidArgumentError, match="inference-only"):
        ob.add_valid_mask_maps_output()
    x = _dev(setup.x)
    for name in ("output/view_ang/view_ang", "output/cen_z_direct/cen_z"):
        with pytest.raises(_lib.InvalidArgumentError, match="no trainable layer"):
            tnet.fully_connected(x, name, False)
    # the layers of monopsr_model_000's set are there and differentiable
    fin = tnet.fc_index["output/lwh/lwh"][1]
    h = torch.randn((B, fin), device="cuda", requires_grad=True)
    y = tnet.fully_connected(h, "output/lwh/lwh", False)
    y.sum().backward()
    assert h.grad is not None and float(h.grad.abs().sum()) > 0
