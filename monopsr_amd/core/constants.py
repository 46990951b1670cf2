"""String keys of the instance path, same values as the reference's core/constants.py:35-77."""
KEY_VALID_MASK_MAPS = 'valid_mask_maps'
KEY_INST_XYZ_MAP_LOCAL = 'inst_xyz_map_local'
KEY_INST_XYZ_MAP_GLOBAL = 'inst_xyz_map_global'
KEY_INST_PROJ_ERR_MAP = 'inst_proj_err_map'
KEY_INST_DEPTH_MAP_GLOBAL = 'inst_depth_map_global'
KEY_INST_XYZ_MAP_GLOBAL_FROM_DEPTH = 'inst_xyz_map_global_from_depth'
KEY_BOX_2D = 'box_2d'
KEY_BOX_3D = 'box_3d'
KEY_PROP_CEN_Z = 'prop_cen_z'
KEY_VIEW_ANG = 'view_ang'
KEY_CEN_X = 'cen_x'
KEY_CEN_Y = 'cen_y'
KEY_CEN_Z = 'cen_z'
KEY_LWH = 'lwh'
KEY_ALPHA = 'alpha'
KEY_ALPHA_BINS = 'alpha_bins'
KEY_ALPHA_REGS = 'alpha_regs'
KEY_CENTROIDS = 'centroids'

NET_IN_RGB_CROP = 'net_in_rgb_crop'
NET_IN_FULL_IMG = 'net_in_full_img'
# Not in the reference: BASELINE configs 2-5 feed the full-image branch as its already cropped + pooled
# (B, 12, 12, 1024) feature map; when this key is present NET_IN_FULL_IMG is not needed.
NET_IN_FULL_IMG_FEATURE_CROP = 'net_in_full_img_feature_crop'

FEATURES_FOR_MAP = 'features_for_map'
FEATURES_FOR_BOX_3D = 'features_for_box_3d'
FEATURES_PROPOSAL_FC_OUT = 'features_proposal_fc_out'
FEATURES_REGRESSION_FC_OUT = 'features_regression_fc_out'

METRIC_EMD = 'metric_emd'
METRIC_CHAMFER = 'metric_chamfer'

# sample-dict keys used by format_predictions / save_predictions (core/constants.py:2-32)
SAMPLE_IMAGE_INPUT = 'sample_image_input'
SAMPLE_NUM_OBJS = 'sample_num_objs'
SAMPLE_LABEL_BOXES_2D = 'sample_label_boxes_2d'
SAMPLE_LABEL_BOXES_3D = 'sample_label_boxes_3d'
SAMPLE_VIEWING_ANGLES_3D = 'sample_viewing_angles_3d'
SAMPLE_LABEL_CLASS_INDICES = 'sample_label_class_indices'
SAMPLE_LABEL_SCORES = 'sample_label_scores'
SAMPLE_CAM_P = 'sample_cam_p'
SAMPLE_NAME = 'sample_name'

OUT_DIR_BOX_2D = 'output_box_2d_dir'
OUT_DIR_BOX_3D = 'output_box_3d_dir'
OUT_DIR_XYZ_MAP_LOCAL = 'output_xyz_map_dir'
