"""Feature extraction of the instance path, mirroring builders/net_builder.py:17-96 of the reference.

Same entry point, arguments and returned dict keys; the graph underneath is libmonopsr_hip.so:
two ResNet-101 trunks (crop, full image) -> feature crop + 2x2 max-pool -> [concat] -> 1x1 squash ->
2x2 max-pool (FEATURES_FOR_BOX_3D) / map decoder (FEATURES_FOR_MAP).
"""
import torch

from monopsr_amd.core import constants
from monopsr_amd.core import device_net as dn


def get_net_config(model_config):
    net_type = model_config.net_type
    return getattr(model_config.net_config, net_type)


def extract_features(model, net_type, model_config, input_dict, is_training):
    """model: object with `device_net` (DeviceNet), `pl_boxes_2d_norm` (B,4), `num_boxes`, `map_roi_size`,
    `is_training`.  input_dict: NET_IN_RGB_CROP (B,h,w,3) and either NET_IN_FULL_IMG (1,H,W,3) or
    NET_IN_FULL_IMG_FEATURE_CROP (B, map_h/4, map_w/4, 1024)."""
    features_dict = {}
    if net_type == 'resnet101_4x_squash':
        return build_resnet101_4x_squash(getattr(model_config, 'net_config', None), net_type, model, input_dict,
                                         features_dict)
    raise ValueError('Invalid net_type', net_type)


def build_resnet101_4x_squash(net_config, net_type, model, input_dict, features_dict):
    net = model.device_net
    crop_img = input_dict[constants.NET_IN_RGB_CROP]

    # BatchNorm: the trunks run on their moving statistics (frozen in the reference too,
    # faster_rcnn_resnet_v1_feature_extractor.py:63,238); the decoder's does so at inference and, in a trainable
    # net built with decoder_bn='batch', uses batch statistics as net_builder.py:78-79,86-87 does when is_training.
    if constants.NET_IN_FULL_IMG_FEATURE_CROP in input_dict:
        crop_img_encoder_out = net.trunk(crop_img, 'crop')
        full_img_feature_crop = input_dict[constants.NET_IN_FULL_IMG_FEATURE_CROP]
    else:
        full_img = input_dict[constants.NET_IN_FULL_IMG]
        half = (model.map_roi_size[0] // 2, model.map_roi_size[1] // 2)

        def full_branch():
            # full-image trunk, then crop and resize + max pool of its feature map (net_builder.py:46-60); a
            # trainable net brings differentiable versions of the two operators (the crop's image gradient feeds
            # the full-image trunk's backward pass)
            crop_and_resize = getattr(net, 'crop_and_resize', dn.crop_and_resize)
            max_pool = getattr(net, 'max_pool', dn.max_pool)
            full_img_encoder_out = net.trunk(full_img, 'full')
            large = crop_and_resize(full_img_encoder_out, model.pl_boxes_2d_norm, None, half)
            return max_pool(large, 2, 2, "VALID")

        # (a trainable net caps the box count, TrainNet.side_stream_max_boxes = 64: training step with both trunks 24.5 vs
        # 26.2 ms at 32 boxes; above, a third busy stream wins or loses with the hardware queues it happens to get)
        few_enough = crop_img.shape[0] <= getattr(net, 'side_stream_max_boxes', 1 << 30)
        if hasattr(net, 'side_stream') and few_enough and torch.cuda.is_available():
            # The two trunks are independent and, at one image / a few dozen boxes, each leaves most CUs idle
            # (M = 6080 and 32*144 rows) -- run the full-image branch on a second HIP stream next to the crop
            # trunk.  (A trainable net with both trunks does the same since r06: autograd runs a node's backward on
            # the stream of its forward, so the backward passes of the two trunks overlap too.)
            if net.side_stream is None:
                net.side_stream = dn.concurrent_stream(crop_img.device)
            main = torch.cuda.current_stream()
            net.side_stream.wait_stream(main)
            with torch.cuda.stream(net.side_stream):
                full_img_feature_crop = full_branch()
            crop_img_encoder_out = net.trunk(crop_img, 'crop')
            main.wait_stream(net.side_stream)
            full_img_feature_crop.record_stream(main)
        else:
            crop_img_encoder_out = net.trunk(crop_img, 'crop')
            full_img_feature_crop = full_branch()

    # concat + 1x1 squash + pool + map decoder in one native call; the xyz-map head (a 3x3 conv on the map
    # features, monopsr_output_builder.py:95-104) rides along and is handed to the output builder
    features_pooled, map_features, xyz = net.squash_decoder(crop_img_encoder_out, full_img_feature_crop,
                                                            tuple(model.map_roi_size))
    features_dict.update({
        constants.FEATURES_FOR_MAP: map_features,
        constants.FEATURES_FOR_BOX_3D: features_pooled,
        '_' + constants.KEY_INST_XYZ_MAP_LOCAL: xyz,
    })
    return features_dict
