"""CPU restatement (PyTorch fp32/fp64 on the host) of MonoPSR's network hot path: crop, ResNet-101 trunk to block3
at output stride 4, squash + map decoder, xyz-map head and the centroid / shape regression heads.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

PARITY UNPINNED.  The arithmetic of this part of the reference lives in TensorFlow 1.8 (slim.conv2d / batch_norm /
fully_connected / max_pool2d, tf.image.crop_and_resize / resize_bilinear), which is neither vendored in
/root/reference nor installable here, and the reference holds no test or fixture for it.
The graph below follows the reference's Python line by line (citations on each function); the TF operator
semantics are restated from TF 1.8's documented kernels.  Each operator is cross-checked against an independent
naive numpy implementation in tests/test_oracle_net.py.

Unlike the product, nothing is folded or fused here: convolution, BatchNorm, bias, activation and residual add are
separate steps in the reference's order.  Tensors are NHWC torch tensors; weights a dict keyed by the reference's
TF variable names (monopsr_amd.core.weights documents the names; the dict is data, not code).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

TRUNK_BLOCKS = (("block1", 64, 3, 1), ("block2", 128, 4, 2), ("block3", 256, 23, 4))


def _t(a, dtype):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.asarray(a)).to(dtype)


# ------------------------------------------------------------------------------------------- TF operators

def tf_conv2d(x, w_hwio, stride=1, rate=1, padding="SAME"):
    """tf.nn.conv2d / atrous conv on NHWC input with an HWIO kernel.  SAME at stride 1 pads (k-1)*rate/2 per side."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    w = w_hwio.permute(3, 2, 0, 1).contiguous()
    xin = x.permute(0, 3, 1, 2)
    if padding == "SAME":
        assert stride == 1
        ph, pw = (kh - 1) * rate // 2, (kw - 1) * rate // 2
        y = F.conv2d(xin, w, None, stride=1, padding=(ph, pw), dilation=rate)
    else:
        y = F.conv2d(xin, w, None, stride=stride, padding=0, dilation=rate)
    return y.permute(0, 2, 3, 1).contiguous()


def tf_batch_norm(x, gamma, beta, mean, var, eps):
    """slim.batch_norm in inference mode: (x - mean) * rsqrt(var + eps) * gamma + beta (gamma optional)."""
    inv = torch.rsqrt(var + eps)
    if gamma is not None:
        inv = inv * gamma
    return x * inv + (beta - mean * inv)


def tf_max_pool(x, k, s, padding):
    """slim.max_pool2d.  SAME: out = ceil(in/s), total pad = max((out-1)*s+k-in, 0), pad_before = total//2
    (so an odd total puts the extra row/column at the bottom/right); padded cells never win."""
    b, h, w, c = x.shape
    if padding == "SAME":
        oh, ow = -(-h // s), -(-w // s)
        th, tw = max((oh - 1) * s + k - h, 0), max((ow - 1) * s + k - w, 0)
        xin = F.pad(x.permute(0, 3, 1, 2), (tw // 2, tw - tw // 2, th // 2, th - th // 2), value=-math.inf)
    else:
        xin = x.permute(0, 3, 1, 2)
    return F.max_pool2d(xin, k, s).permute(0, 2, 3, 1).contiguous()


def tf_resize_bilinear(x, out_h, out_w, align_corners):
    """tf.image.resize_bilinear, TF-1.8 kernel (no half-pixel centres): src = dst * scale with
    scale = (in-1)/(out-1) if align_corners and out > 1 else in/out; lower = floor(src), upper = min(lower+1, in-1);
    top = tl + (tr-tl)*xl ; bottom = bl + (br-bl)*xl ; out = top + (bottom-top)*yl."""
    b, h, w, c = x.shape
    if (out_h, out_w) == (h, w):
        return x  # tf.image.resize_images returns its input when the size is unchanged

    def axis(n_in, n_out):
        scale = (n_in - 1) / (n_out - 1) if (align_corners and n_out > 1) else n_in / n_out
        # TF computes the source coordinate in float32
        src = (torch.arange(n_out, dtype=torch.float32) * np.float32(scale))
        lo = torch.floor(src).to(torch.int64)
        hi = torch.clamp(lo + 1, max=n_in - 1)
        return lo, hi, (src - lo.to(torch.float32)).to(x.dtype)

    y0, y1, yl = axis(h, out_h)
    x0, x1, xl = axis(w, out_w)
    xl = xl.view(1, 1, -1, 1)
    yl = yl.view(1, -1, 1, 1)
    rows0, rows1 = x[:, y0], x[:, y1]
    top = rows0[:, :, x0] + (rows0[:, :, x1] - rows0[:, :, x0]) * xl
    bot = rows1[:, :, x0] + (rows1[:, :, x1] - rows1[:, :, x0]) * xl
    return top + (bot - top) * yl


def tf_crop_and_resize(image, boxes, box_ind, crop_h, crop_w, extrapolation_value=0.0):
    """tf.image.crop_and_resize (bilinear).  boxes [y1,x1,y2,x2] normalised to [0,1] over (H-1),(W-1);
    in_y = y1*(H-1) + i*(y2-y1)*(H-1)/(crop_h-1) (crop_h > 1) else 0.5*(y1+y2)*(H-1); samples outside
    [0, H-1] x [0, W-1] take extrapolation_value; top = floor, bottom = ceil, lerp = in - floor.
    Coordinates are float32 as in the TF kernel."""
    n, h, w, c = image.shape
    boxes = _t(boxes, torch.float32).to(torch.float32)
    nb = boxes.shape[0]
    out = torch.full((nb, crop_h, crop_w, c), float(extrapolation_value), dtype=image.dtype)
    f32 = np.float32
    for bi in range(nb):
        y1, x1, y2, x2 = [f32(v) for v in boxes[bi].tolist()]
        img = image[int(box_ind[bi])]
        hs = (y2 - y1) * f32(h - 1) / f32(crop_h - 1) if crop_h > 1 else f32(0)
        ws = (x2 - x1) * f32(w - 1) / f32(crop_w - 1) if crop_w > 1 else f32(0)
        for i in range(crop_h):
            in_y = y1 * f32(h - 1) + f32(i) * hs if crop_h > 1 else f32(0.5) * (y1 + y2) * f32(h - 1)
            if in_y < 0 or in_y > h - 1:
                continue
            top, bot = int(math.floor(in_y)), int(math.ceil(in_y))
            yl = f32(in_y - f32(top))
            xs = np.arange(crop_w, dtype=np.float32)
            in_x = (x1 * f32(w - 1) + xs * ws) if crop_w > 1 else np.full(1, f32(0.5) * (x1 + x2) * f32(w - 1), f32)
            ok = (in_x >= 0) & (in_x <= w - 1)
            if not ok.any():
                continue
            cols = np.nonzero(ok)[0]
            fx = in_x[ok]
            left = np.floor(fx).astype(np.int64)
            right = np.ceil(fx).astype(np.int64)
            xl = torch.from_numpy((fx - left.astype(np.float32)).astype(np.float32)).to(image.dtype).view(-1, 1)
            tl, tr = img[top, left], img[top, right]
            bl, br = img[bot, left], img[bot, right]
            t = tl + (tr - tl) * xl
            b_ = bl + (br - bl) * xl
            out[bi, i, cols] = t + (b_ - t) * float(yl)
    return out


# ------------------------------------------------------------------------------------------- trunk

def _conv_bn(x, W, name, rate, relu, eps, stride=1, padding="SAME"):
    dt = x.dtype
    y = tf_conv2d(x, _t(W[name + "/weights"], dt), stride=stride, rate=rate, padding=padding)
    g = W.get(name + "/BatchNorm/gamma")
    y = tf_batch_norm(y, _t(g, dt) if g is not None else None, _t(W[name + "/BatchNorm/beta"], dt),
                      _t(W[name + "/BatchNorm/moving_mean"], dt), _t(W[name + "/BatchNorm/moving_variance"], dt), eps)
    return torch.relu(y) if relu else y


def bottleneck(x, W, prefix, depth, rate):
    """resnet_v1.bottleneck (resnet_v1.py:79-139) with stride 1 and atrous `rate` on conv2, as stack_blocks_dense
    runs every unit once the target output stride is reached (resnet_utils.py:194-196)."""
    if x.shape[-1] == depth:
        shortcut = x  # subsample(inputs, 1) is the identity (resnet_utils.py:59-74)
    else:
        shortcut = _conv_bn(x, W, prefix + "/shortcut", 1, False, 1e-5)
    r = _conv_bn(x, W, prefix + "/conv1", 1, True, 1e-5)
    r = _conv_bn(r, W, prefix + "/conv2", rate, True, 1e-5)
    r = _conv_bn(r, W, prefix + "/conv3", 1, False, 1e-5)
    return torch.relu(shortcut + r)


def resnet101_block3(img, W, scope, collect=None):
    """FasterRCNNResnet101FeatureExtractor._extract_proposal_features with output_stride=4
    (faster_rcnn_resnet_v1_feature_extractor.py:197-245; resnet_v1.py:221-236,310-330).
    Root: explicit pad 3 + 7x7/2 VALID conv + BN + ReLU (resnet_utils.py:115-122), 3x3/2 SAME max-pool; then
    block1 (rate 1), block2 (rate 2), block3 (rate 4), every unit at stride 1; returns block3 activations."""
    dt = img.dtype
    x = F.pad(img.permute(0, 3, 1, 2), (3, 3, 3, 3)).permute(0, 2, 3, 1)
    x = _conv_bn(x, W, scope + "/conv1", 1, True, 1e-5, stride=2, padding="VALID")
    x = tf_max_pool(x, 3, 2, "SAME")
    if collect is not None:
        collect["pool1"] = x
    for block, depth_b, units, rate in TRUNK_BLOCKS:
        scale = W[scope + "/%s/unit_1/bottleneck_v1/conv1/weights" % block].shape[3]  # narrow test copies
        for u in range(1, units + 1):
            x = bottleneck(x, W, "%s/%s/unit_%d/bottleneck_v1" % (scope, block, u), scale * 4, rate)
        if collect is not None:
            collect[block] = x
    return x.to(dt)


# ------------------------------------------------------------------------------------------- squash / decoder / xyz

def squash_decoder(crop_feat, full_feat, W, map_h, map_w):
    """net_builder.py:62-89 (+ the xyz-map head monopsr_output_builder.py:95-104).
    -> features_for_box_3d, features_for_map, inst_xyz_map_local."""
    dt = crop_feat.dtype
    x = torch.cat([crop_feat, full_feat], dim=3)
    sq = tf_conv2d(x, _t(W["squash/1x1_conv/weights"], dt)) + _t(W["squash/1x1_conv/biases"], dt)
    sq = torch.relu(sq)
    feat_box = tf_max_pool(sq, 2, 2, "VALID")
    y = tf_resize_bilinear(sq, map_h // 2, map_w // 2, True)
    y = _conv_bn(y, W, "map_decoder/conv2/conv2_1", 1, True, 1e-3)
    y = _conv_bn(y, W, "map_decoder/conv2/conv2_2", 1, True, 1e-3)
    y = tf_resize_bilinear(y, map_h, map_w, True)
    y = _conv_bn(y, W, "map_decoder/conv3/conv3_1", 1, True, 1e-3)
    feat_map = _conv_bn(y, W, "map_decoder/conv3/conv3_2", 1, True, 1e-3)
    n = "output/inst_xyz_map_local/inst_xyz_map_local"
    xyz = tf_conv2d(feat_map, _t(W[n + "/weights"], dt)) + _t(W[n + "/biases"], dt)
    return feat_box, feat_map, xyz


# ------------------------------------------------------------------------------------------- heads

def _fc(x, W, name, relu):
    y = x @ _t(W[name + "/weights"], x.dtype) + _t(W[name + "/biases"], x.dtype)
    return torch.relu(y) if relu else y


def heads(feat_box3d, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, W, image_shape=(320, 1216),
          num_classes=1, num_alpha_bins=12, max_depth=45.0, cen_y_norm=1.666754, cen_y_class_offset=0.0648):
    """monopsr_output_builder.py:126-302,407-438,457-488,509-661 wired as monopsr_model.py:320-413 with the
    model_000 output_config (lwh offset, alpha dc, view_ang est, cen_y/cen_z offset, cen_x from_view_ang_and_z).
    Dropout keep_prob 1.0 is the identity.  Returns the reference's output_dict entries."""
    dt = feat_box3d.dtype
    B = feat_box3d.shape[0]
    flat = feat_box3d.reshape(B, -1)  # slim.flatten of NHWC
    boxes_2d, cam_p = _t(boxes_2d, dt).to(dt), _t(cam_p, dt).to(dt).reshape(3, 4)
    view = _t(view_angs, dt).to(dt).reshape(B, 1)
    mean_lwh = _t(mean_lwh, dt).to(dt)
    z_off = _t(cen_z_offset, dt).to(dt).reshape(B)
    class_idx = _t(class_idx, torch.int64).reshape(B).to(torch.int64)
    # obj_utils.tf_boxes_2d_ij_fmt (obj_utils.py:1016-1034)
    cu, cv = cam_p[0, 2], cam_p[1, 2]
    coords = boxes_2d - torch.stack([cv, cu, cv, cu])
    heights = (boxes_2d[:, 2] - boxes_2d[:, 0]).unsqueeze(1)
    heights_n = heights / image_shape[0]
    hh, hw = image_shape[0] / 2.0, image_shape[1] / 2.0
    coords_n = coords / torch.tensor([hh, hw, hh, hw], dtype=dt)
    onehot = torch.zeros((B, num_classes), dtype=dt)  # tf.one_hot: out-of-range index -> all off
    for i in range(B):
        if 0 <= int(class_idx[i]) < num_classes:
            onehot[i, int(class_idx[i])] = 1.0
    cam_n = cam_p.reshape(1, 12) / torch.tensor([1000.0, 1.0, 1000.0, 100.0, 1.0, 1000.0, 1000.0, 1.0, 1.0, 1.0,
                                                 1.0, 1.0], dtype=dt)
    p = "output/proposal_fc/proposal_fc/"
    img_fc = _fc(flat, W, p + "img_fc", True)
    f = torch.cat([img_fc, coords_n, heights_n, view, onehot, cam_n.repeat(B, 1)], dim=1)
    f = _fc(f, W, p + "fc0", True)
    prop_feat = _fc(f, W, p + "fc1", True)
    out = {}
    lwh_offs = _fc(prop_feat, W, "output/lwh/lwh", False)
    out["lwh_offs"] = lwh_offs
    out["lwh"] = mean_lwh + lwh_offs
    alpha = _fc(prop_feat, W, "output/alpha", False)
    out["alpha_bins"], out["alpha_regs"] = alpha[:, :num_alpha_bins], alpha[:, num_alpha_bins:2 * num_alpha_bins]
    out["view_ang"] = view
    # get_prop_cen_z (:407-431), tf_est_y_from_box_2d_and_depth (instance_utils.py:907-953)
    focal = cam_p[0, 0]
    prop_z = (focal * out["lwh"][:, 2] / (boxes_2d[:, 2] - boxes_2d[:, 0]) + z_off).unsqueeze(1)
    out["prop_cen_z"] = prop_z
    centre_v = ((boxes_2d[:, 2] + boxes_2d[:, 0]) / 2.0 - cv).unsqueeze(1)
    prop_y = centre_v * (prop_z / focal) - cen_y_class_offset
    r = "output/regression_fc/regression_fc/"
    img_fc_r = _fc(flat, W, r + "img_fc", True)
    g = torch.cat([img_fc_r, coords_n, heights_n, view, onehot, lwh_offs, out["alpha_bins"], out["alpha_regs"],
                   prop_y / cen_y_norm, prop_z / max_depth], dim=1)
    g = _fc(g, W, r + "fc0", True)
    reg_feat = _fc(g, W, r + "fc1", True)
    out["cen_y_offs"] = _fc(reg_feat, W, "output/cen_y/cen_y", False)
    out["cen_y"] = prop_y + out["cen_y_offs"]
    out["cen_z_offs"] = _fc(reg_feat, W, "output/cen_z_offs/cen_z", False)
    out["cen_z"] = prop_z + out["cen_z_offs"]
    out["cen_x"] = out["cen_z"] * torch.tan(view) + (-cam_p[0, 3] / cam_p[0, 0])
    out["centroids"] = torch.cat([out["cen_x"], out["cen_y"], out["cen_z"]], dim=1)
    return out


def instance_path(rgb_crops, full_feat_crop, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, W,
                  scope="FirstStageFeatureExtractor_crop/resnet_v1_101", map_size=(48, 48), dtype=torch.float32):
    """Proposal crops in -> output_dict (centroids + N x 3 local point cloud ...) out, for BASELINE configs 1-3:
    the full-image branch enters as its (B, 12, 12, 1024) cropped+pooled feature map."""
    x = _t(rgb_crops, dtype).to(dtype)
    crop_feat = resnet101_block3(x, W, scope)
    feat_box, feat_map, xyz = squash_decoder(crop_feat, _t(full_feat_crop, dtype).to(dtype), W, *map_size)
    out = heads(feat_box, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, W)
    out.update({"crop_feat": crop_feat, "features_for_box_3d": feat_box, "features_for_map": feat_map,
                "inst_xyz_map_local": xyz})
    return out


KITTI_CHANNEL_MEANS = (92.8403, 97.7996, 93.5843)


def full_image_path(rgb_image, boxes_2d, boxes_2d_norm, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, W,
                    image_input_shape=(320, 1216), resized_full_img_shape=(160, 608), img_roi_size=(48, 48),
                    map_size=(48, 48), dtype=torch.float32):
    """The whole per-image instance path as MonoPSRModel.build wires it (monopsr_model.py:130-131, 222-237;
    img_preprocessor.py:12-35; net_builder.py:44-60): mean-subtract + resize the image, crop the proposals,
    run both trunks, crop + pool the full-image features, then squash / decoder / heads."""
    img = _t(rgb_image, dtype).to(dtype).unsqueeze(0) - torch.tensor(KITTI_CHANNEL_MEANS, dtype=dtype)
    pre = tf_resize_bilinear(img, image_input_shape[0], image_input_shape[1], False)
    nb = len(boxes_2d_norm)
    crops = tf_crop_and_resize(pre, boxes_2d_norm, np.zeros(nb, np.int32), img_roi_size[0], img_roi_size[1])
    full = tf_resize_bilinear(pre, resized_full_img_shape[0], resized_full_img_shape[1], True)
    crop_feat = resnet101_block3(crops, W, "FirstStageFeatureExtractor_crop/resnet_v1_101")
    full_feat = resnet101_block3(full, W, "FirstStageFeatureExtractor_full/resnet_v1_101")
    large = tf_crop_and_resize(full_feat, boxes_2d_norm, np.zeros(nb, np.int32), map_size[0] // 2, map_size[1] // 2)
    full_crop = tf_max_pool(large, 2, 2, "VALID")
    feat_box, feat_map, xyz = squash_decoder(crop_feat, full_crop, W, *map_size)
    out = heads(feat_box, boxes_2d, cam_p, view_angs, class_idx, mean_lwh, cen_z_offset, W,
                image_shape=image_input_shape)
    out.update({"rgb_crops": crops, "crop_feat": crop_feat, "full_feat": full_feat, "full_feat_crop": full_crop,
                "features_for_box_3d": feat_box, "features_for_map": feat_map, "inst_xyz_map_local": xyz})
    return out
