"""Harness for the instance path of MonoPSRModel (core/models/monopsr/monopsr_model.py of the reference):
`build` = crop wiring :222-233 + net_builder + head wiring :248-462 (incl. the train/val global maps),
`loss` = :554-958, `format_predictions` / `save_predictions` = :960-1102, `evaluate_predictions` = the Chamfer /
EMD metrics :1112-1170.  The feed-dict machinery of the TF1 session and checkpoint restore are replaced by the
`sample` dict passed to `build` (placeholder names with the `pl_` prefix dropped) and core/checkpoint_utils.py.
"""
import os

import numpy as np
import torch

from monopsr_amd.builders import loss_builder
from monopsr_amd.datasets.kitti import instance_utils

from monopsr_amd.builders import net_builder
from monopsr_amd.core import constants
from monopsr_amd.core import device_net as dn
from monopsr_amd.core.img_preprocessor import ImgPreprocessor
from monopsr_amd.core.models.monopsr import monopsr_output_builder
from monopsr_amd.tf_ops.approxmatch import tf_approxmatch
from monopsr_amd.tf_ops.nn_distance import tf_nndistance


class MonoPSRModel:

    def __init__(self, model_config, dataset_config, device_net, train_val_test='test', classes_name='Car',
                 fused_heads=True):
        self.model_config = model_config
        self.dataset_config = dataset_config
        self.device_net = device_net
        self.train_val_test = train_val_test
        self.is_training = train_val_test == 'train'
        self.classes_name = classes_name
        self.fused_heads = fused_heads
        self.num_boxes = dataset_config.num_boxes
        self.image_input_shape = list(model_config.image_input_shape)
        self.img_roi_size = list(model_config.img_roi_size)
        self.map_roi_size = list(model_config.map_roi_size)
        self.resized_full_img_shape = list(model_config.resized_full_img_shape)
        self.mean_sub_type = model_config.mean_sub_type
        self.net_type = model_config.net_type
        self.output_config = model_config.output_config
        self.output_types = monopsr_output_builder.MonoPSROutputBuilder.get_output_types_list(self.output_config)
        self.depth_range = list(dataset_config.obj_filter_config.depth_range)
        self.img_preprocessor = ImgPreprocessor()
        self.centroid_type = model_config.get('centroid_type', 'middle')
        self.rotate_view = bool(model_config.get('rotate_view', True))
        self.post_process_cen_x = bool(model_config.get('post_process_cen_x', True))
        self.num_alpha_bins = dataset_config.num_alpha_bins

    # ------------------------------------------------------------------ build
    def build(self, sample):
        """sample keys (GPU tensors): either `rgb_image` (H,W,3) [+ `boxes_2d_norm` (B,4)] for the full path, or
        `rgb_image_crops` (B,h,w,3) + `full_img_feature_crop` (B,h/4,w/4,1024) (BASELINE configs 2-5);
        `boxes_2d` (B,4) pixels [y1,x1,y2,x2], `cam_p` (3,4), `est_view_angs` (B), `class_indices` (B,1),
        `mean_lwh` (B,3), `prop_cen_z_offset` (B).  Returns (output_dict, features_dict)."""
        B = sample['boxes_2d'].shape[0]
        self.num_boxes = B
        self.pl_boxes_2d_norm = sample.get('boxes_2d_norm')
        input_dict = {}
        if 'rgb_image_crops' in sample:
            input_dict[constants.NET_IN_RGB_CROP] = sample['rgb_image_crops']
            input_dict[constants.NET_IN_FULL_IMG_FEATURE_CROP] = sample['full_img_feature_crop']
        else:
            rgb_image_batched = sample['rgb_image'].unsqueeze(0)
            self.img_preprocessed = self.img_preprocessor.preprocess_input(
                rgb_image_batched, self.image_input_shape, mean_sub_type=self.mean_sub_type)
            # monopsr_model.py:222-226 / :228-233
            input_dict[constants.NET_IN_RGB_CROP] = dn.crop_and_resize(
                self.img_preprocessed, self.pl_boxes_2d_norm, None, tuple(self.img_roi_size))
            input_dict[constants.NET_IN_FULL_IMG] = dn.resize_bilinear(
                self.img_preprocessed, tuple(self.resized_full_img_shape), align_corners=True)
        features_dict = net_builder.extract_features(self, self.net_type, self.model_config, input_dict,
                                                     self.is_training)
        return self.build_outputs(features_dict, sample), features_dict

    def build_batch(self, samples):
        """N images with their boxes in ONE pass (inference, fused heads): `samples` = a list of the dicts `build` takes
        (`rgb_image` (H_i,W_i,3), `boxes_2d_norm`, `boxes_2d`, `cam_p`, `est_view_angs`, `class_indices`, `mean_lwh`,
        `prop_cen_z_offset`; box counts and image sizes may differ per image, and an image may have NO boxes: its dict then
        holds empty tensors).  The reference's step is one image
        (monopsr_model.py:222-237, configs/monopsr_model_000.yaml:14-17); here every image is preprocessed as there,
        then all of them go through DeviceNet.forward_images: one full-image trunk call with batch N, one crop-trunk /
        decoder / heads call with all boxes.  Returns a list of per-image output dicts (views into the batch
        tensors), each equal to what `build(sample)` returns for that image."""
        if not self.fused_heads or self.train_val_test != 'test':
            raise ValueError("build_batch is the inference path (train_val_test='test', fused_heads=True)")
        if len(samples) == 0:
            return []
        dev = samples[0]['boxes_2d'].device
        pre = [self.img_preprocessor.preprocess_input(s['rgb_image'].unsqueeze(0), self.image_input_shape,
                                                      mean_sub_type=self.mean_sub_type) for s in samples]
        images = torch.cat(pre, 0)
        counts = [int(s['boxes_2d'].shape[0]) for s in samples]
        if sum(counts) == 0:  # no box in any image: nothing to run (every output is per box)
            nb = self.dataset_config.num_alpha_bins
            z = lambda *shape: torch.zeros((0,) + shape, dtype=torch.float32, device=dev)
            empty = {constants.KEY_INST_XYZ_MAP_LOCAL: z(self.map_roi_size[0], self.map_roi_size[1], 3),
                     "lwh": z(3), "lwh_offs": z(3), "alpha_bins": z(nb), "alpha_regs": z(nb), "centroids": z(3)}
            empty.update({k: z(1) for k in ("prop_cen_z", "cen_y", "cen_y_offs", "cen_z", "cen_z_offs", "cen_x",
                                            "view_ang")})
            return [dict(empty, **{constants.SAMPLE_LABEL_CLASS_INDICES: s['class_indices']}) for s in samples]
        box_ind = torch.cat([torch.full((c,), i, dtype=torch.int32, device=dev) for i, c in enumerate(counts)])
        # (explicit trailing sizes: reshape(0, -1) of an image WITHOUT boxes is ambiguous and raises)
        cat = lambda k, w: torch.cat([s[k].reshape(c, w) for s, c in zip(samples, counts)], 0)
        cam_p = torch.stack([s['cam_p'].reshape(3, 4) for s in samples], 0)
        xyz, out = self.device_net.forward_images(
            images, cat('boxes_2d_norm', 4), box_ind, cat('boxes_2d', 4), cam_p, cat('est_view_angs', 1).reshape(-1),
            cat('class_indices', 1), cat('mean_lwh', 3), cat('prop_cen_z_offset', 1).reshape(-1),
            img_roi_size=tuple(self.img_roi_size), map_roi_size=tuple(self.map_roi_size),
            resized_full_img_shape=tuple(self.resized_full_img_shape),
            image_shape=self.image_input_shape, max_depth=self.depth_range[1],
            num_classes=len(self.dataset_config.classes), num_alpha_bins=self.dataset_config.num_alpha_bins,
            cen_y_class_offset=monopsr_output_builder.CEN_Y_CLASS_OFFSET[(self.classes_name, 'kitti')])
        out[constants.KEY_INST_XYZ_MAP_LOCAL] = xyz
        results, lo = [], 0
        for s, c in zip(samples, counts):
            o = {k: v[lo:lo + c] for k, v in out.items()}
            o[constants.SAMPLE_LABEL_CLASS_INDICES] = s['class_indices']
            if s.get('gt_valid_mask_maps') is not None:
                o[constants.KEY_VALID_MASK_MAPS] = s['gt_valid_mask_maps']
            results.append(o)
            lo += c
        return results

    def build_outputs(self, features_dict, sample):
        boxes_2d = sample['boxes_2d']
        cam_p = sample['cam_p'].reshape(3, 4)
        est_view_angs = sample['est_view_angs'].reshape(-1, 1)
        if self.fused_heads:
            if self.train_val_test != 'test':
                raise ValueError("fused_heads is the inference fast path; build train/val models with fused_heads=False")
            out = self.device_net.heads_fwd(
                features_dict[constants.FEATURES_FOR_BOX_3D], boxes_2d, cam_p, sample['est_view_angs'],
                sample['class_indices'], sample['mean_lwh'], sample['prop_cen_z_offset'],
                image_shape=self.image_input_shape, max_depth=self.depth_range[1],
                num_classes=len(self.dataset_config.classes), num_alpha_bins=self.dataset_config.num_alpha_bins,
                cen_y_class_offset=monopsr_output_builder.CEN_Y_CLASS_OFFSET[(self.classes_name, 'kitti')])
            out[constants.KEY_INST_XYZ_MAP_LOCAL] = features_dict['_' + constants.KEY_INST_XYZ_MAP_LOCAL]
            out[constants.SAMPLE_LABEL_CLASS_INDICES] = sample['class_indices']
            if sample.get('gt_valid_mask_maps') is not None:  # :315-318: the fed masks stand in for predicted ones
                out[constants.KEY_VALID_MASK_MAPS] = sample['gt_valid_mask_maps']
            return out

        # method-by-method wiring of monopsr_model.py:248-462
        is_tv = self.train_val_test in ['train', 'val']
        gt_lwh = gt_alpha_dc = gt_cen_z = gt_view_angs = gt_cen_y = gt_centroids = None
        gt_valid_mask_maps = sample.get('gt_valid_mask_maps')
        gt_inst_xyz_maps_global = sample.get('gt_inst_xyz_maps_global')
        if is_tv:
            # :261-281 ground truth from the (N,7) 3-D boxes [x,y,z,l,w,h,ry]; 'middle' moves y to the box centre
            tf_boxes_3d = sample['boxes_3d']
            gt_cen_y = tf_boxes_3d[:, 1:2] - tf_boxes_3d[:, 5:6] / 2 if self.centroid_type == 'middle' \
                else tf_boxes_3d[:, 1:2]
            gt_cen_z = tf_boxes_3d[:, 2:3]
            gt_centroids = torch.cat([tf_boxes_3d[:, 0:1], gt_cen_y, gt_cen_z], dim=1)
            gt_lwh = tf_boxes_3d[:, 3:6]
            gt_alpha_dc = [sample['gt_alpha_bins'].reshape(-1, 1), sample['gt_alpha_regs']]
            gt_view_angs = sample['gt_view_angs'].reshape(-1, 1)
        builder = monopsr_output_builder.MonoPSROutputBuilder(
            self.output_config, self.model_config, self.dataset_config, features_dict, self.num_boxes,
            self.map_roi_size, cam_p, train_val_test=self.train_val_test, device_net=self.device_net)
        output_dict = builder.get_output_dict()
        if constants.KEY_INST_XYZ_MAP_LOCAL in self.output_types:
            builder.add_inst_xyz_maps_local(gt_inst_xyz_maps_local=sample.get('gt_inst_xyz_maps_local'))
        # :308-318 valid mask maps: predicted by their own head when configured (model 000 predicts none), else the
        # ground-truth masks stand in
        if gt_valid_mask_maps is not None:
            builder._gt_dict.add_unique_to_dict({constants.KEY_VALID_MASK_MAPS: gt_valid_mask_maps})
        if constants.KEY_VALID_MASK_MAPS in self.output_types:
            builder.add_valid_mask_maps_output()
        elif gt_valid_mask_maps is not None:
            output_dict.add_unique_to_dict({constants.KEY_VALID_MASK_MAPS: gt_valid_mask_maps})
        builder.add_proposal_fc_features(boxes_2d=boxes_2d, view_angs=est_view_angs,
                                         class_indices=sample['class_indices'], image_shape=self.image_input_shape)
        init_est_fc_features = builder.get_proposal_fc_features()
        if constants.KEY_LWH in self.output_types:
            builder.add_lwh_output(features_to_use=init_est_fc_features, est_lwh=sample['mean_lwh'], gt_lwh=gt_lwh)
        if constants.KEY_ALPHA in self.output_types:
            builder.add_alpha_output(features_to_use=init_est_fc_features, gt_alpha=sample.get('gt_alpha'),
                                     gt_alpha_dc=gt_alpha_dc)
        if constants.KEY_VIEW_ANG in self.output_types:
            builder.add_view_ang_output(output_key=constants.KEY_VIEW_ANG, features_in=init_est_fc_features,
                                        est_view_angs=est_view_angs, gt_view_angs=gt_view_angs)
        prop_cen_z = builder.get_prop_cen_z(boxes_2d, sample['prop_cen_z_offset'])
        prop_cen_y = builder.get_prop_cen_y(boxes_2d, prop_cen_z, self.classes_name)
        builder.add_regression_fc_features(
            boxes_2d=boxes_2d, view_angs=est_view_angs, class_indices=sample['class_indices'],
            image_shape=self.image_input_shape, est_lwh_off=output_dict[constants.KEY_LWH + '_offs'],
            est_alpha_bins=output_dict[constants.KEY_ALPHA_BINS], est_alpha_regs=output_dict[constants.KEY_ALPHA_REGS],
            prop_cen_y=prop_cen_y, prop_cen_z=prop_cen_z, max_depth=self.depth_range[1])
        regression_fc_features = builder.get_regression_fc_features()
        builder.add_cen_y_output(output_key=constants.KEY_CEN_Y, features_in=regression_fc_features,
                                 prop_cen_y=prop_cen_y, gt_cen_y=gt_cen_y)
        builder.add_cen_z_output(output_key=constants.KEY_CEN_Z, features_in=regression_fc_features,
                                 prop_cen_z=prop_cen_z, gt_cen_z=gt_cen_z)
        if constants.KEY_CEN_X in self.output_types:
            builder.add_cen_x_output(output_key=constants.KEY_CEN_X, pred_cen_z=output_dict[constants.KEY_CEN_Z],
                                     pred_view_angs=output_dict[constants.KEY_VIEW_ANG])
        if constants.KEY_CENTROIDS in self.output_types:
            builder.add_centroids_output(output_key=constants.KEY_CENTROIDS,
                                         pred_cen_x=output_dict[constants.KEY_CEN_X],
                                         pred_cen_y=output_dict[constants.KEY_CEN_Y],
                                         pred_cen_z=output_dict[constants.KEY_CEN_Z], gt_centroids=gt_centroids)
        # # # Global maps (:414-462), train / val only # # #
        if is_tv:
            if constants.KEY_INST_XYZ_MAP_GLOBAL in self.output_types:
                # the map is placed with the GROUND-TRUTH viewing angle at the predicted depth / height
                pred_cen_y = output_dict[constants.KEY_CEN_Y]
                pred_cen_z = output_dict[constants.KEY_CEN_Z]
                x_offset = -cam_p[0, 3] / cam_p[0, 0]
                proj_gt_cen_x = pred_cen_z * torch.tan(gt_view_angs) + x_offset
                proj_pred_cen = torch.cat([proj_gt_cen_x, pred_cen_y, pred_cen_z], dim=1)
                pred_inst_xyz_maps_global = builder.get_inst_xyz_map_global(
                    pred_inst_xyz_maps_local=output_dict[constants.KEY_INST_XYZ_MAP_LOCAL],
                    pred_view_angs=gt_view_angs, pred_centroids=proj_pred_cen)
                proj_err_norm, _ = builder.get_proj_err_maps_norm(
                    pred_inst_xyz_map_global=pred_inst_xyz_maps_global, pred_boxes_2d=boxes_2d,
                    valid_mask_maps=gt_valid_mask_maps)
                output_dict.add_unique_to_dict({'proj_err_norm': proj_err_norm})
            if constants.KEY_INST_DEPTH_MAP_GLOBAL in self.output_types and \
                    constants.KEY_INST_XYZ_MAP_LOCAL in self.output_types:
                builder.add_inst_depth_maps_global(
                    pred_inst_depth_maps_local=output_dict[constants.KEY_INST_XYZ_MAP_LOCAL][:, :, :, 2:3],
                    gt_inst_depth_maps_global=gt_inst_xyz_maps_global[:, :, :, 2:3],
                    rotate_view=self.rotate_view, box_2d=boxes_2d)
        self.gt_dict = builder.get_gt_dict()
        out = builder.get_output()
        out[constants.SAMPLE_LABEL_CLASS_INDICES] = sample['class_indices']
        return out

    # ------------------------------------------------------------------ loss (:554-958)
    def loss(self, output_dict, gt_dict, gt_alpha_valid_bins=None):
        """Total training loss of the built output set -> (losses_dict, total_loss).  gt_dict is `self.gt_dict`
        after build(); gt_alpha_valid_bins (B, num_alpha_bins) is the reference's pl_gt_alpha_valid_bins."""
        loss_config = self.model_config.loss_config
        num_boxes = float(self.num_boxes)
        dev = output_dict[constants.KEY_LWH].device
        loss_mask_ones = torch.ones((1, self.num_boxes, 1), dtype=torch.float32, device=dev)
        losses_dict = {}
        total_loss = 0

        def box_term(pred, gt, cfg_key, mask=loss_mask_ones):
            return loss_builder.add_loss_tensor(loss_config, cfg_key, pred.unsqueeze(0), gt.unsqueeze(0),
                                                mask=mask).sum() / num_boxes

        if constants.KEY_INST_XYZ_MAP_LOCAL in self.output_types:
            v = loss_builder.add_loss_tensor(
                loss_config, constants.KEY_INST_XYZ_MAP_LOCAL, output_dict[constants.KEY_INST_XYZ_MAP_LOCAL],
                gt_dict[constants.KEY_INST_XYZ_MAP_LOCAL], mask=gt_dict[constants.KEY_VALID_MASK_MAPS]) / num_boxes
            losses_dict[constants.KEY_INST_XYZ_MAP_LOCAL] = v
            total_loss = total_loss + v
        if constants.KEY_VALID_MASK_MAPS in self.output_types:
            # :597-620 label-smoothed targets, every pixel weighted 1, mean over the pixels, summed over the instances
            gt_mask = gt_dict[constants.KEY_VALID_MASK_MAPS]
            mask_map_loss = loss_builder.add_loss_tensor(
                loss_config, constants.KEY_VALID_MASK_MAPS, output_dict[constants.KEY_VALID_MASK_MAPS],
                gt_mask * 0.998 + 0.001, torch.ones_like(gt_mask))
            v = (mask_map_loss.sum(dim=(1, 2)) / float(self.map_roi_size[0] * self.map_roi_size[1])).sum()
            losses_dict[constants.KEY_VALID_MASK_MAPS] = v
            total_loss = total_loss + v
        if constants.KEY_LWH in self.output_types:
            k = constants.KEY_LWH + '_offs'
            losses_dict[k] = box_term(output_dict[k], gt_dict[k], constants.KEY_LWH)
            total_loss = total_loss + losses_dict[k]
        if constants.KEY_ALPHA in self.output_types:
            alpha_type = self.output_config.alpha
            gt_bins = gt_dict[constants.KEY_ALPHA_BINS].reshape(-1).long()
            rows = torch.arange(self.num_boxes, device=dev)
            if alpha_type in ('dc', 'dc_rotation'):  # :655-710
                eps = getattr(loss_config, constants.KEY_ALPHA + '_cls')[2]
                one_hot = torch.full((self.num_boxes, self.num_alpha_bins), eps / self.dataset_config.num_alpha_bins,
                                     dtype=torch.float32, device=dev)
                # (scatter_: index_put with index tensors cannot be captured into a HIP graph on this build)
                one_hot.scatter_(1, gt_bins.reshape(-1, 1), 1.0 - eps)
                bins_loss = box_term(output_dict[constants.KEY_ALPHA_BINS], one_hot, constants.KEY_ALPHA + '_cls')
                reg_loss = box_term(output_dict[constants.KEY_ALPHA_REGS], gt_dict[constants.KEY_ALPHA_REGS],
                                    constants.KEY_ALPHA + '_reg', mask=gt_alpha_valid_bins.unsqueeze(0).float())
                losses_dict[constants.KEY_ALPHA_BINS] = bins_loss
                losses_dict[constants.KEY_ALPHA_REGS] = reg_loss
                total_loss = total_loss + (bins_loss + reg_loss)
            elif alpha_type == 'prob':  # :712-752: hard one-hot targets, the temperature softmax, alpha itself regressed
                one_hot = torch.zeros((self.num_boxes, self.num_alpha_bins), dtype=torch.float32, device=dev)
                one_hot.scatter_(1, gt_bins.reshape(-1, 1), 1.0)
                bins_loss = box_term(output_dict[constants.KEY_ALPHA_BINS], one_hot, constants.KEY_ALPHA + '_cls_temp')
                reg_loss = box_term(output_dict[constants.KEY_ALPHA], gt_dict[constants.KEY_ALPHA],
                                    constants.KEY_ALPHA + '_reg')
                losses_dict[constants.KEY_ALPHA_BINS] = bins_loss
                losses_dict[constants.KEY_ALPHA] = reg_loss
                total_loss = total_loss + (bins_loss + reg_loss)
            elif alpha_type != 'gt':  # 'gt': nothing to learn
                raise ValueError('Invalid output_type', alpha_type)
        for key in (constants.KEY_CEN_Z, constants.KEY_VIEW_ANG, constants.KEY_CEN_Y):
            if key in self.output_types and getattr(self.output_config, key) == 'offset':
                k = key + '_offs'
                losses_dict[k] = box_term(output_dict[k], gt_dict[k], key)
                total_loss = total_loss + losses_dict[k]
        if constants.KEY_INST_XYZ_MAP_GLOBAL in self.output_types and 'proj_err_norm' in output_dict:
            proj_err_norm = output_dict['proj_err_norm'].reshape(1, -1, 1)
            v = loss_builder.add_loss_tensor(loss_config, constants.KEY_INST_XYZ_MAP_GLOBAL, proj_err_norm,
                                             torch.zeros_like(proj_err_norm), mask=loss_mask_ones)
            losses_dict['proj_err'] = v
            total_loss = total_loss + v
        if constants.KEY_INST_DEPTH_MAP_GLOBAL in self.output_types and \
                constants.KEY_INST_DEPTH_MAP_GLOBAL in output_dict:
            v = loss_builder.add_loss_tensor(
                loss_config, constants.KEY_INST_DEPTH_MAP_GLOBAL, output_dict[constants.KEY_INST_DEPTH_MAP_GLOBAL],
                gt_dict[constants.KEY_INST_DEPTH_MAP_GLOBAL], mask=gt_dict[constants.KEY_VALID_MASK_MAPS]) / num_boxes
            losses_dict[constants.KEY_INST_DEPTH_MAP_GLOBAL] = v
            total_loss = total_loss + v
        return losses_dict, total_loss

    # ------------------------------------------------------------------ predictions (:960-1102)
    def format_predictions(self, output_types, output_dict, sample_dict):
        """-> pred_dict with KEY_VALID_MASK_MAPS, KEY_INST_XYZ_MAP_LOCAL (masked), KEY_BOX_3D (n,9) and KEY_BOX_2D
        (n,7) as numpy arrays, n = sample_dict[SAMPLE_NUM_OBJS].  sample_dict keys: SAMPLE_NUM_OBJS, SAMPLE_CAM_P,
        SAMPLE_LABEL_SCORES, SAMPLE_LABEL_BOXES_2D, SAMPLE_IMAGE_INPUT (only its shape is used) or 'image_shape'."""
        num_objs = int(sample_dict[constants.SAMPLE_NUM_OBJS])
        pred_dict = {}
        dev = output_dict[constants.KEY_LWH].device
        valid_mask_maps = (output_dict[constants.KEY_VALID_MASK_MAPS][0:num_objs] > 0.0).float()
        pred_dict[constants.KEY_VALID_MASK_MAPS] = valid_mask_maps.cpu().numpy()
        if constants.KEY_INST_XYZ_MAP_LOCAL in output_types:
            masked = output_dict[constants.KEY_INST_XYZ_MAP_LOCAL][0:num_objs] * valid_mask_maps
            pred_dict[constants.KEY_INST_XYZ_MAP_LOCAL] = masked.cpu().numpy()
        if constants.KEY_CENTROIDS in output_types:
            if 'image_shape' in sample_dict:
                img_shape = sample_dict['image_shape']
            else:
                img_shape = sample_dict[constants.SAMPLE_IMAGE_INPUT].shape
            as_dev = lambda a, dt=torch.float32: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a,
                                                                 dtype=dt, device=dev)
            box_3d, box_2d = instance_utils.format_boxes(
                output_dict[constants.KEY_LWH], output_dict[constants.KEY_VIEW_ANG],
                output_dict[constants.KEY_ALPHA_BINS], output_dict[constants.KEY_ALPHA_REGS],
                output_dict[constants.KEY_CENTROIDS], as_dev(sample_dict[constants.SAMPLE_LABEL_BOXES_2D]),
                as_dev(sample_dict[constants.SAMPLE_LABEL_SCORES]),
                as_dev(output_dict[constants.SAMPLE_LABEL_CLASS_INDICES], torch.int32),
                as_dev(sample_dict[constants.SAMPLE_CAM_P]), img_shape, centroid_type=self.centroid_type,
                post_process_cen_x=self.post_process_cen_x)
            pred_dict[constants.KEY_BOX_3D] = box_3d[0:num_objs].cpu().numpy()
            pred_dict[constants.KEY_BOX_2D] = box_2d[0:num_objs].cpu().numpy()
        return pred_dict

    def save_predictions(self, sample_name, predictions, sample_dict, output_dirs):
        """:1073-1102: xyz maps as float16 .npy, boxes as '%0.5f' text (mask PNGs need cv2: not written)."""
        predictions = self.format_predictions(self.output_types, predictions, sample_dict)
        if constants.KEY_INST_XYZ_MAP_LOCAL in self.output_types:
            out_dir = output_dirs[constants.OUT_DIR_XYZ_MAP_LOCAL]
            np.save(os.path.join(out_dir, '{}.npy'.format(sample_name)),
                    predictions[constants.KEY_INST_XYZ_MAP_LOCAL].astype(np.float16))
        if constants.KEY_CENTROIDS in self.output_types:
            np.savetxt(os.path.join(output_dirs[constants.OUT_DIR_BOX_3D], '{}.txt'.format(sample_name)),
                       predictions[constants.KEY_BOX_3D], fmt='%0.5f')
            np.savetxt(os.path.join(output_dirs[constants.OUT_DIR_BOX_2D], '{}.txt'.format(sample_name)),
                       predictions[constants.KEY_BOX_2D], fmt='%0.5f')
        return predictions

    # ------------------------------------------------------------------ metrics (:1112-1170)
    def evaluate_predictions(self, prediction_dict, gt_dict, num_objs=None):
        """gt_dict: KEY_INST_XYZ_MAP_LOCAL (B,h,w,3) and KEY_VALID_MASK_MAPS (B,h,w,1).  Returns METRIC_EMD and
        METRIC_CHAMFER per object, each divided by its number of valid pixels."""
        metrics_dict = {}
        pred = prediction_dict[constants.KEY_INST_XYZ_MAP_LOCAL]
        gt = gt_dict[constants.KEY_INST_XYZ_MAP_LOCAL]
        mask = gt_dict[constants.KEY_VALID_MASK_MAPS]
        B = pred.shape[0]
        num_objs = B if num_objs is None else int(num_objs)
        valid_pred_inst_points = (pred * mask).reshape(B, -1, 3)
        valid_gt_points = (gt * mask).reshape(B, -1, 3)
        num_valid_pixels = mask[0:num_objs].sum(dim=(1, 2, 3))
        with torch.no_grad():
            # approx_match + match_cost (monopsr_model.py:1143-1149), fused: no match tensor
            all_distances = tf_approxmatch.emd_loss_fwd_bwd(valid_pred_inst_points, valid_gt_points,
                                                            want_grads=False)[0]
            metrics_dict[constants.METRIC_EMD] = all_distances[0:num_objs] / num_valid_pixels
            dist1, _, dist2, _ = tf_nndistance.nn_distance(valid_pred_inst_points, valid_gt_points)
            all_chamfer_dists = dist1.sum(dim=1) + dist2.sum(dim=1)
            metrics_dict[constants.METRIC_CHAMFER] = all_chamfer_dists[0:num_objs] / num_valid_pixels
        return metrics_dict
