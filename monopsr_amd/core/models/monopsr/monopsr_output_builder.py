"""Output heads of the instance path, mirroring core/models/monopsr/monopsr_output_builder.py:37-661.

Method names, arguments, output_dict keys and the order of operations are the reference's; every fully connected
layer runs through the HIP fp32-MFMA GEMM (DeviceNet.fully_connected), the per-box scalar algebra (a few flops
per box) is plain tensor arithmetic on the GPU.  The fused native fast path for the same graph is
mpsr_heads_fwd (DeviceNet.heads_fwd); tests check that both agree.
Every output type the reference's methods accept is implemented for the FORWARD path on a DeviceNet ('offset' /
'est' / 'gt' / 'direct', alpha 'dc' / 'dc_rotation' / 'prob' / 'gt', the predicted valid-mask head);
monopsr_model_000.yaml uses the first of each, and that set is what trains: on a TrainNet (core/train_net.py) a
variant that needs a layer outside monopsr_model_000's set raises InvalidArgumentError instead of running without
gradients (tests/test_output_variants_gpu.py).  The global maps (:663-772) run through the geometry kernels (datasets/kitti/instance_utils.py); box
rescoring (:805-860) is fused into mpsr_format_boxes (MonoPSRModel.format_predictions).
"""
import math

import numpy as np
import torch

from monopsr_amd.core import constants
from monopsr_amd.core import device_net as dn
from monopsr_amd.datasets.kitti import instance_utils

PROP_CEN_Y_NORM = 1.666754  # monopsr_output_builder.py:246
CEN_Y_CLASS_OFFSET = {('Car', 'kitti'): 0.0648, ('Car', 'mscnn'): 0.0655, ('Pedestrian', 'kitti'): 0.0145,
                      ('Pedestrian', 'mscnn'): 0.0142, ('Cyclist', 'kitti'): 0.0239, ('Cyclist', 'mscnn'): 0.0239}
CAM_P_NORM = [1000.0, 1.0, 1000.0, 100.0, 1.0, 1000.0, 1000.0, 1.0, 1.0, 1.0, 1.0, 1.0]


class UniqueKeyDict:
    def __init__(self, init_entries=None):
        self.dict = init_entries if init_entries is not None else {}

    def __getitem__(self, key):
        return self.dict[key]

    def add_unique_to_dict(self, new_entries):
        for key, value in new_entries.items():
            if self.dict.get(key, None) is None:
                self.dict.update({key: value})
            else:
                raise ValueError('Key {} already exists in output_dict'.format(key))


def tf_boxes_2d_ij_fmt(boxes_2d, cam_p):
    """datasets/kitti/obj_utils.py:1016-1034: box coordinates relative to the principal point."""
    centre_u, centre_v = cam_p[0, 2], cam_p[1, 2]
    return boxes_2d - torch.stack([centre_v, centre_u, centre_v, centre_u])


def tf_est_y_from_box_2d_and_depth(cam_p, box_2d, depth, class_str=None, trend_data='kitti'):
    """datasets/kitti/instance_utils.py:907-953."""
    focal_length, centre_v = cam_p[0, 0], cam_p[1, 2]
    box_2d_centre_v = ((box_2d[:, 2] + box_2d[:, 0]) / 2. - centre_v).unsqueeze(1)
    cen_y_mid_estimate = box_2d_centre_v * (depth / focal_length)
    if (class_str, trend_data) not in CEN_Y_CLASS_OFFSET:
        raise ValueError('Invalid class_str', class_str)
    return cen_y_mid_estimate - CEN_Y_CLASS_OFFSET[(class_str, trend_data)]


class MonoPSROutputBuilder:

    def __init__(self, output_config, model_config, dataset_config, features_dict, num_boxes, map_roi_size, cam_p,
                 train_val_test, device_net=None):
        self.output_config = output_config.__dict__ if hasattr(output_config, '__dict__') else dict(output_config)
        self.output_types = MonoPSROutputBuilder.get_output_types_list(output_config)
        self.model_config = model_config
        self.dataset_config = dataset_config
        self._output_dict = UniqueKeyDict()
        self._features_dict = UniqueKeyDict(features_dict)
        self._gt_dict = UniqueKeyDict()
        self.num_boxes = num_boxes
        self.map_roi_size = map_roi_size
        self.cam_p = cam_p.reshape(3, 4)
        self.features_for_map = features_dict[constants.FEATURES_FOR_MAP]
        self.features_for_box_3d = features_dict[constants.FEATURES_FOR_BOX_3D]
        self.train_val_test = train_val_test
        self.is_training = train_val_test == 'train'
        self.is_train_or_val = train_val_test in ['train', 'val']
        self.net = device_net

    @staticmethod
    def get_output_types_list(output_config):
        d = output_config.__dict__ if hasattr(output_config, '__dict__') else dict(output_config)
        return sorted([k for k in d.keys() if not k.startswith('__')])

    def get_output_dict(self):
        return self._output_dict

    def get_output(self):
        return self._output_dict.dict

    def get_features_dict(self):
        return self._features_dict.dict

    def get_gt_dict(self):
        return self._gt_dict.dict

    # ------------------------------------------------------------------ local xyz map (:95-108)
    def add_inst_xyz_maps_local(self, gt_inst_xyz_maps_local):
        output_key = constants.KEY_INST_XYZ_MAP_LOCAL
        output = self._features_dict.dict.get('_' + output_key)
        if output is None:
            raise ValueError('features_dict holds no xyz map; build the features with net_builder.extract_features')
        self._output_dict.add_unique_to_dict({output_key: output})
        if self.is_train_or_val:
            self._gt_dict.add_unique_to_dict({output_key: gt_inst_xyz_maps_local})

    # ------------------------------------------------------------------ predicted valid mask (:110-120)
    def add_valid_mask_maps_output(self):
        """3x3 convolution of the map features to one logit per pixel (no activation), variables
        output/valid_mask_maps/valid_mask_maps/{weights,biases}; through the narrow-N 3x3 HIP kernel."""
        from monopsr_amd.core import device_net
        output_key = constants.KEY_VALID_MASK_MAPS
        name = 'output/' + output_key + '/' + output_key
        if self.net is not None and not hasattr(self.net, '_weights'):
            from monopsr_amd import _lib
            raise _lib.InvalidArgumentError(
                'the predicted valid-mask head is inference-only: %s holds no trainable layer for %r (its '
                'convolution has no gradient path and its weights are outside the flat parameter buffer); build the '
                'model on a DeviceNet, or drop valid_mask_maps from output_config for training'
                % (type(self.net).__name__, name))
        if self.net is None or name + '/weights' not in self.net._weights:
            raise ValueError('no weights for the predicted valid-mask head', name)
        if name not in self.net._fc_cache:
            w = np.asarray(self.net._weights[name + '/weights'], np.float32)  # HWIO (3, 3, cin, 1)
            w_ok = np.ascontiguousarray(w.transpose(3, 0, 1, 2).reshape(w.shape[3], -1))
            self.net._fc_cache[name] = (torch.from_numpy(w_ok).to(self.net.device),
                                        torch.from_numpy(np.asarray(self.net._weights[name + '/biases'],
                                                                    np.float32)).to(self.net.device))
        w_ok, bias = self.net._fc_cache[name]
        output = device_net.conv2d(self.features_for_map, w_ok, bias, None, 3, 3, 1, False)
        self._output_dict.add_unique_to_dict({output_key: output})

    # ------------------------------------------------------------------ shared scalar features
    def _box_features(self, boxes_2d, class_indices, image_shape):
        box_2d_coords = tf_boxes_2d_ij_fmt(boxes_2d, self.cam_p)
        box_2d_heights = (boxes_2d[:, 2] - boxes_2d[:, 0]).unsqueeze(1)
        box_2d_heights_norm = box_2d_heights / image_shape[0]
        half_img_height, half_img_width = image_shape[0] / 2.0, image_shape[1] / 2.0
        # (scalar divisors and an arithmetic one-hot: no host -> device copy, no data-dependent shape -- the step can be
        # captured into a HIP graph)
        c = box_2d_coords
        box_2d_coords_norm = torch.cat([c[:, 0:1] / half_img_height, c[:, 1:2] / half_img_width,
                                        c[:, 2:3] / half_img_height, c[:, 3:4] / half_img_width], dim=1)
        num_classes = len(self.dataset_config.classes)
        idx = class_indices.reshape(-1).to(torch.int64)
        # tf.one_hot: out-of-range index -> all off
        one_hot = (idx.unsqueeze(1) == torch.arange(num_classes, device=boxes_2d.device).unsqueeze(0)).to(torch.float32)
        return box_2d_coords_norm, box_2d_heights_norm, one_hot

    # ------------------------------------------------------------------ proposal FC stack (:126-194)
    def get_proposal_fc_features(self):
        return self._features_dict[constants.FEATURES_PROPOSAL_FC_OUT]

    def add_proposal_fc_features(self, boxes_2d, view_angs, class_indices, image_shape):
        p = 'output/proposal_fc/proposal_fc/'
        flat_img_features = self.features_for_box_3d.reshape(self.features_for_box_3d.shape[0], -1)
        coords_norm, heights_norm, one_hot = self._box_features(boxes_2d, class_indices, image_shape)
        img_fc = self.net.fully_connected(flat_img_features, p + 'img_fc', True)
        cam_p_normalized = self.cam_p.reshape(1, -1) / dn.device_constant([CAM_P_NORM], self.cam_p.device)
        cam_p_tiled = cam_p_normalized.repeat(self.num_boxes, 1)
        features_concat = torch.cat([img_fc, coords_norm, heights_norm, view_angs, one_hot, cam_p_tiled], dim=1)
        fc_drop = features_concat
        for fc_idx, _ in enumerate(self.model_config.proposal_fc_layers.layer_sizes):
            fc_drop = self.net.fully_connected(fc_drop, p + 'fc{}'.format(fc_idx), True)  # dropout keep 1.0
        self._features_dict.add_unique_to_dict({constants.FEATURES_PROPOSAL_FC_OUT: fc_drop})

    # ------------------------------------------------------------------ regression FC stack (:200-274)
    def get_regression_fc_features(self):
        return self.get_features_dict()[constants.FEATURES_REGRESSION_FC_OUT]

    def add_regression_fc_features(self, boxes_2d, view_angs, class_indices, image_shape, est_lwh_off,
                                   est_alpha_bins, est_alpha_regs, prop_cen_y, prop_cen_z, max_depth):
        r = 'output/regression_fc/regression_fc/'
        flat_img_features = self.features_for_box_3d.reshape(self.features_for_box_3d.shape[0], -1)
        coords_norm, heights_norm, one_hot = self._box_features(boxes_2d, class_indices, image_shape)
        prop_cen_y_norm = prop_cen_y / PROP_CEN_Y_NORM
        prop_cen_z_norm = prop_cen_z / max_depth
        img_fc = self.net.fully_connected(flat_img_features, r + 'img_fc', True)
        features_concat = torch.cat([img_fc, coords_norm, heights_norm, view_angs, one_hot, est_lwh_off,
                                     est_alpha_bins, est_alpha_regs, prop_cen_y_norm, prop_cen_z_norm], dim=1)
        fc_drop = features_concat
        for fc_idx, _ in enumerate(self.model_config.regression_fc_layers.layer_sizes):
            fc_drop = self.net.fully_connected(fc_drop, r + 'fc{}'.format(fc_idx), True)
        self._features_dict.add_unique_to_dict({constants.FEATURES_REGRESSION_FC_OUT: fc_drop})

    # ------------------------------------------------------------------ outputs
    def add_alpha_output(self, features_to_use, gt_alpha, gt_alpha_dc):
        """:276-393.  'dc': bins + one regression per bin; 'dc_rotation': bins + a (cos, sin) pair per bin,
        L2-normalised, regression = atan2; 'prob': bins only, alpha = angle of the softmax-weighted bin centres;
        'gt': ground truth passed through."""
        output_key = constants.KEY_ALPHA
        output_type = self.output_config[output_key]
        num_alpha_bins = self.dataset_config.num_alpha_bins
        if output_type == 'dc':
            orientation_outputs = self.net.fully_connected(features_to_use, 'output/alpha', False)
            outputs = {constants.KEY_ALPHA_BINS: orientation_outputs[:, 0:num_alpha_bins],
                       constants.KEY_ALPHA_REGS: orientation_outputs[:, num_alpha_bins:num_alpha_bins * 2]}
        elif output_type == 'dc_rotation':
            orientation_outputs = self.net.fully_connected(features_to_use, 'output/alpha', False)
            comp = orientation_outputs[:, num_alpha_bins:num_alpha_bins * 3].reshape(self.num_boxes, num_alpha_bins, 2)
            # tf.nn.l2_normalize(axis=2, epsilon=1e-12): x * rsqrt(max(sum x^2, eps))
            comp = comp * torch.rsqrt(torch.clamp((comp * comp).sum(2, keepdim=True), min=1e-12))
            outputs = {constants.KEY_ALPHA_BINS: orientation_outputs[:, 0:num_alpha_bins],
                       constants.KEY_ALPHA_REGS: torch.atan2(comp[:, :, 1], comp[:, :, 0])}
        elif output_type == 'prob':
            pred_alpha_bins = self.net.fully_connected(features_to_use, 'output/alpha', False)
            angle_per_half_bin = 2 * math.pi / num_alpha_bins / 2
            centres = np.linspace(angle_per_half_bin, 2 * math.pi - angle_per_half_bin, num_alpha_bins)
            bin_centers_comp = dn.device_constant(np.stack((np.cos(centres), np.sin(centres)), axis=1),
                                                  pred_alpha_bins.device)
            comp = torch.softmax(pred_alpha_bins, dim=1) @ bin_centers_comp
            outputs = {constants.KEY_ALPHA_BINS: pred_alpha_bins,
                       constants.KEY_ALPHA: torch.atan2(comp[:, 1], comp[:, 0]).unsqueeze(1)}
        elif output_type == 'gt':
            outputs = {constants.KEY_ALPHA_BINS: gt_alpha_dc[0], constants.KEY_ALPHA_REGS: gt_alpha_dc[1]}
        else:
            raise ValueError('Invalid output_type', output_type)
        self._output_dict.add_unique_to_dict(outputs)
        if self.is_train_or_val and gt_alpha_dc is not None:
            if output_type == 'prob':
                self._gt_dict.add_unique_to_dict({constants.KEY_ALPHA_BINS: gt_alpha_dc[0],
                                                  constants.KEY_ALPHA: gt_alpha})
            else:
                self._gt_dict.add_unique_to_dict({constants.KEY_ALPHA_BINS: gt_alpha_dc[0],
                                                  constants.KEY_ALPHA_REGS: gt_alpha_dc[1]})

    def add_lwh_output(self, features_to_use, est_lwh, gt_lwh):
        output_key = constants.KEY_LWH
        output_type = self.output_config[output_key]
        if output_type == 'offset':
            pred_dim_offsets = self.net.fully_connected(features_to_use, 'output/lwh/lwh', False)
            pred_lwh = est_lwh + pred_dim_offsets
        elif output_type == 'est':
            pred_dim_offsets = est_lwh
            pred_lwh = est_lwh
        elif output_type == 'gt':
            pred_dim_offsets = gt_lwh - est_lwh
            pred_lwh = gt_lwh
        else:
            raise ValueError('Invalid output_type', output_type)
        self._output_dict.add_unique_to_dict({output_key + '_offs': pred_dim_offsets, output_key: pred_lwh})
        if self.is_train_or_val and gt_lwh is not None:
            self._gt_dict.add_unique_to_dict({output_key: gt_lwh, output_key + '_offs': gt_lwh - pred_lwh})
        return pred_lwh

    def add_view_ang_output(self, output_key, features_in, est_view_angs, gt_view_angs):
        """:509-549.  'offset': one FC output added to the 2-D viewing angle (variables
        output/<key>/<key>); 'est': the 2-D viewing angle itself; 'gt': ground truth."""
        output_type = self.output_config[output_key]
        if output_type == 'offset':
            pred_view_ang_offsets = self.net.fully_connected(features_in, 'output/%s/%s' % (output_key, output_key),
                                                             False)
            pred_view_angs = est_view_angs + pred_view_ang_offsets
        elif output_type == 'est':
            pred_view_angs = est_view_angs
            pred_view_ang_offsets = torch.zeros((), dtype=torch.float32, device=est_view_angs.device)
        elif output_type == 'gt':
            pred_view_ang_offsets = gt_view_angs - est_view_angs
            pred_view_angs = gt_view_angs
        else:
            raise ValueError('Invalid output_type', output_type)
        self._output_dict.add_unique_to_dict({output_key + '_offs': pred_view_ang_offsets,
                                              output_key: pred_view_angs})
        if self.is_train_or_val and gt_view_angs is not None:
            self._gt_dict.add_unique_to_dict({output_key + '_offs': gt_view_angs - est_view_angs,
                                              output_key: gt_view_angs})

    def get_prop_cen_z(self, boxes_2d, offset):
        f = self.cam_p[0, 0]
        est_obj_h = self.get_output_dict()[constants.KEY_LWH][:, 2]
        boxes_2d_h = boxes_2d[:, 2] - boxes_2d[:, 0]
        prop_cen_z = (f * est_obj_h / boxes_2d_h + offset).unsqueeze(1)
        self._output_dict.add_unique_to_dict({constants.KEY_PROP_CEN_Z: prop_cen_z})
        return prop_cen_z

    def get_prop_cen_y(self, boxes_2d, depth, class_name):
        return tf_est_y_from_box_2d_and_depth(self.cam_p, boxes_2d, depth, class_name, trend_data='kitti')

    def add_cen_z_output(self, output_key, features_in, prop_cen_z, gt_cen_z):
        """:441-507.  'offset': FC output added to the proposal depth (scope cen_z_offs); 'direct': the FC output
        is the depth (scope cen_z_direct, no offset entry)."""
        output_type = self.output_config[output_key]
        assert prop_cen_z.shape[1] == 1
        if output_type == 'offset':
            pred_cen_z_offsets = self.net.fully_connected(features_in, 'output/cen_z_offs/' + output_key, False)
            pred_cen_z = prop_cen_z + pred_cen_z_offsets
            self._output_dict.add_unique_to_dict({output_key + '_offs': pred_cen_z_offsets, output_key: pred_cen_z})
            if self.is_train_or_val and gt_cen_z is not None:
                self._gt_dict.add_unique_to_dict({output_key: gt_cen_z, output_key + '_offs': gt_cen_z - prop_cen_z})
        elif output_type == 'direct':
            pred_cen_z = self.net.fully_connected(features_in, 'output/cen_z_direct/' + output_key, False)
            self._output_dict.add_unique_to_dict({output_key: pred_cen_z})
            if self.is_train_or_val and gt_cen_z is not None:
                self._gt_dict.add_unique_to_dict({output_key: gt_cen_z})
        else:
            raise ValueError('Invalid output_type', output_type)

    def add_cen_y_output(self, output_key, features_in, prop_cen_y, gt_cen_y):
        """:573-609.  'offset' / 'gt'; the reference's 'est' branch leaves the offset undefined and fails with a
        NameError when the outputs are collected -- same here."""
        output_type = self.output_config[output_key]
        if output_type == 'offset':
            pred_cen_y_offsets = self.net.fully_connected(features_in, 'output/cen_y/' + output_key, False)
            pred_cen_y = prop_cen_y + pred_cen_y_offsets
        elif output_type == 'est':
            raise NameError("name 'pred_cen_y_offsets' is not defined")  # monopsr_output_builder.py:587-598
        elif output_type == 'gt':
            pred_cen_y_offsets = gt_cen_y - prop_cen_y
            pred_cen_y = gt_cen_y
        else:
            raise ValueError('Invalid output_type', output_type)
        self._output_dict.add_unique_to_dict({output_key + '_offs': pred_cen_y_offsets, output_key: pred_cen_y})
        if self.is_train_or_val and gt_cen_y is not None:
            self._gt_dict.add_unique_to_dict({output_key + '_offs': gt_cen_y - prop_cen_y, output_key: gt_cen_y})

    def add_cen_x_output(self, output_key, pred_cen_z, pred_view_angs):
        output_type = self.output_config[output_key]
        if output_type != 'from_view_ang_and_z':
            raise ValueError('Invalid output_type', output_type)
        cam2_pred_cen_x = pred_cen_z * torch.tan(pred_view_angs)
        x_offset = -self.cam_p[0, 3] / self.cam_p[0, 0]
        self._output_dict.add_unique_to_dict({output_key: cam2_pred_cen_x + x_offset})

    def add_centroids_output(self, output_key, pred_cen_x, pred_cen_y, pred_cen_z, gt_centroids):
        self._output_dict.add_unique_to_dict({output_key: torch.cat([pred_cen_x, pred_cen_y, pred_cen_z], dim=1)})
        if self.is_train_or_val and gt_centroids is not None:
            self._gt_dict.add_unique_to_dict({output_key: gt_centroids})

    # ------------------------------------------------------------------ global maps (:663-772)
    def get_inst_xyz_map_global(self, pred_inst_xyz_maps_local, pred_view_angs, pred_centroids):
        """Global point cloud map from the local xyz map (instance_utils.tf_inst_xyz_map_local_to_global)."""
        return instance_utils.tf_inst_xyz_map_local_to_global(pred_inst_xyz_maps_local, self.map_roi_size,
                                                              pred_view_angs, pred_centroids)

    def get_proj_err_maps_norm(self, pred_inst_xyz_map_global, pred_boxes_2d, valid_mask_maps, debug=False):
        """:681-746 -> (proj_err_norm (N,), proj_err_debug_dict).  The debug dict holds the normalised error map
        only when `debug` (the reference's other debug entries are intermediate tensors of its TF graph)."""
        proj_err_norm, maps = instance_utils.proj_err_maps_norm(pred_inst_xyz_map_global, pred_boxes_2d, self.cam_p,
                                                                valid_mask_maps, want_maps=debug)
        proj_err_debug_dict = {'pred_inst_xyz_map_global': pred_inst_xyz_map_global}
        if debug:
            proj_err_debug_dict['proj_err_map_norm'] = maps
        return proj_err_norm, proj_err_debug_dict

    def add_inst_depth_maps_global(self, pred_inst_depth_maps_local, gt_inst_depth_maps_global, rotate_view, box_2d):
        """:748-772: local depth + predicted centroid depth (+ view-normalisation offset)."""
        output_key = constants.KEY_INST_DEPTH_MAP_GLOBAL
        pred_cen_z = self._output_dict[constants.KEY_CEN_Z]
        inst_view_ang = self._output_dict[constants.KEY_VIEW_ANG]
        pred = instance_utils.tf_inst_depth_map_local_to_global(
            pred_inst_depth_maps_local, pred_cen_z, box_2d, inst_view_ang, self.map_roi_size, self.cam_p, rotate_view)
        self._output_dict.add_unique_to_dict({output_key: pred})
        if self.is_train_or_val:
            self._gt_dict.add_unique_to_dict({output_key: gt_inst_depth_maps_global})

    def add_inst_xyz_maps_global_from_depth(self, pred_inst_depth_map_global, boxes_2d, gt_inst_xyz_maps_global):
        """:774-803: back-project each instance's global depth map (N,h,w,1) through the pixel-centre grid of its
        2-D box (datasets/kitti/depth_map_utils.py:161-236, tf_depth_patch_to_pc_map: x = (u - cu) d / f,
        y = (v - cv) d / f, z = d).  The reference turns each (3, h, w) result into (1, h, w, 3) with a RESHAPE, not
        a transpose, and no caller in the reference reaches this method; the reshape is kept as written."""
        output_key = constants.KEY_INST_XYZ_MAP_GLOBAL_FROM_DEPTH
        h, w = self.map_roi_size
        d = pred_inst_depth_map_global.reshape(-1, h, w)
        n = d.shape[0]
        y1, x1, y2, x2 = [boxes_2d[:n, i].reshape(n, 1) for i in range(4)]
        # roi_size[0] counts the x pixels and roi_size[1] the y pixels in the reference (square maps in practice)
        nx, ny = self.map_roi_size[0], self.map_roi_size[1]
        half_w, half_h = (x2 - x1) / nx / 2.0, (y2 - y1) / ny / 2.0
        tx = torch.linspace(0.0, 1.0, nx, device=d.device).reshape(1, nx)
        ty = torch.linspace(0.0, 1.0, ny, device=d.device).reshape(1, ny)
        xs = (x1 + half_w) + ((x2 - half_w) - (x1 + half_w)) * tx  # (n, nx)
        ys = (y1 + half_h) + ((y2 - half_h) - (y1 + half_h)) * ty  # (n, ny)
        ratio = d / self.cam_p[0, 0]
        x = (xs.reshape(n, 1, nx) - self.cam_p[0, 2]) * ratio
        y = (ys.reshape(n, ny, 1) - self.cam_p[1, 2]) * ratio
        pc_map = torch.stack((x, y, d), dim=1)  # (n, 3, h, w)
        output = pc_map.reshape(n, h, w, 3)
        self._output_dict.add_unique_to_dict({output_key: output})
        if self.is_train_or_val:
            self._gt_dict.add_unique_to_dict({output_key: gt_inst_xyz_maps_global})
