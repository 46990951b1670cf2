"""String keys of the instance path, same values as the reference's core/constants.py:35-77."""
KEY_VALID_MASK_MAPS = 'valid_mask_maps'
KEY_INST_XYZ_MAP_LOCAL = 'inst_xyz_map_local'
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
