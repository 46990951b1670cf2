"""CPU restatement (numpy) of MonoPSR's per-box geometry after the heads: local -> global instance maps, the
projection-error map, the global depth map, and the post-processing that turns head outputs into KITTI boxes.

TEST INFRASTRUCTURE ONLY: imported by tests/ and __graft_entry__.smoke(); never by the product path.

Pinning.  The reference modules these functions follow import TensorFlow / cv2 at module level and cannot be
imported here, so pinning is against the golden vectors the reference's OWN tests hold for them, re-expressed in
tests/test_oracle_geometry.py:
  * instance_utils_test.py:12-25  (expected-projection grid: pixel centres of box [0,10,10,20] at roi 10x10),
  * instance_utils_test.py:52-73  (tf map transform == the numpy point transform),
  * transform_utils_test.py:9-37  (R_y(90 deg) = [[0,0,1],[0,1,0],[-1,0,0]], translation column),
  * orientation_encoder_test.py   (bin centres / wrap to [-pi, pi]).
Functions with no reference test (projection error, global depth map, cen-x post-processing, box scoring) are
PARITY UNPINNED: they follow the cited lines op by op, in the reference's operand order, and are cross-checked in
fp64 against closed forms where one exists.

Every function takes `dtype` (np.float32 to mirror the TF graph, np.float64 to budget rounding).
"""
import numpy as np


def _a(x, dtype):
    return np.asarray(x, dtype=dtype)


def tf_linspace(start, stop, num, dtype=np.float32):
    """tf.linspace (LinSpace kernel): start + i * (stop - start) / (num - 1), evaluated in `dtype`."""
    start, stop = _a(start, dtype), _a(stop, dtype)
    if num == 1:
        return start[..., None]
    step = (stop - start) / dtype(num - 1)
    return start[..., None] + step[..., None] * np.arange(num, dtype=dtype)


# ---------------------------------------------------------------------------------- transforms / projection

def get_tr_mat_batch(ry, t, dtype=np.float32):
    """transform_utils.py:69-108 tf_get_tr_mat_batch: (rot_mat, t_mat), each (N,4,4); rotation about +y."""
    ry = _a(ry, dtype).reshape(-1)
    t = _a(t, dtype).reshape(-1, 3)
    n = ry.shape[0]
    c, s = np.cos(ry), np.sin(ry)
    rot = np.zeros((n, 4, 4), dtype)
    rot[:, 0, 0], rot[:, 0, 2], rot[:, 1, 1], rot[:, 2, 0], rot[:, 2, 2], rot[:, 3, 3] = c, s, 1, -s, c, 1
    tm = np.tile(np.eye(4, dtype=dtype), (n, 1, 1))
    tm[:, :3, 3] = t
    return rot, tm


def inst_xyz_map_local_to_global(xyz_local, view_angs, centroids, dtype=np.float32):
    """instance_utils.py:567-602 tf_inst_xyz_map_local_to_global: rotate every map point about y by the viewing
    angle, then translate by the centroid.  (N,H,W,3) -> (N,H,W,3)."""
    x = _a(xyz_local, dtype)
    n, h, w, _ = x.shape
    rot, tm = get_tr_mat_batch(view_angs, centroids, dtype)
    pc = np.concatenate([x.reshape(n, h * w, 3).transpose(0, 2, 1), np.ones((n, 1, h * w), dtype)], 1)
    out = np.matmul(tm, np.matmul(rot, pc))
    return out[:, :3].transpose(0, 2, 1).reshape(n, h, w, 3).astype(dtype)


def inst_points_local_to_global(points, view_ang, centroid, dtype=np.float64):
    """instance_utils.py:540-564: the single-instance numpy form the reference's test compares the map op to."""
    return inst_xyz_map_local_to_global(_a(points, dtype).reshape(1, 1, -1, 3), [view_ang], [centroid],
                                        dtype).reshape(-1, 3)


def project_pc_to_image(pc, cam_p, dtype=np.float64):
    """calib_utils.py:245-260 / :263-280: (..., 3, N) points -> (..., 2, N) pixel coordinates [u, v]."""
    pc = _a(pc, dtype)
    pad = np.concatenate([pc, np.ones(pc.shape[:-2] + (1, pc.shape[-1]), dtype)], -2)
    uvw = np.matmul(_a(cam_p, dtype).reshape(3, 4), pad)
    return (uvw[..., 0:2, :] / uvw[..., 2:3, :]).astype(dtype)


def get_exp_proj_uv_map(box_2d, roi_size, round_box_2d=False, use_pixel_centres=False, dtype=np.float64):
    """instance_utils.py:683-735 (numpy form): where an evenly spaced roi grid of the 2-D box lands, (H,W,2)."""
    b = _a(box_2d, dtype)
    if round_box_2d:
        b = np.round(b)
    v1, u1, v2, u2 = b
    roi_h, roi_w = roi_size
    du, dv = (u2 - u1) / roi_w, (v2 - v1) / roi_h
    if use_pixel_centres:
        gu = np.linspace(u1 + du / 2, u2 - du / 2, roi_w)
        gv = np.linspace(v1 + dv / 2, v2 - dv / 2, roi_h)
    else:
        gu = np.linspace(u1, u2 - du, roi_w)
        gv = np.linspace(v1, v2 - dv, roi_h)
    uu, vv = np.meshgrid(gu, gv)
    return np.dstack([uu, vv]).astype(dtype)


def tf_get_exp_proj_uv_map(boxes_2d, roi_size, dtype=np.float32):
    """instance_utils.py:738-788 with the defaults the model uses (no rounding, pixel centres): (N,H,W,2),
    channel 0 = u (varies along W), channel 1 = v (varies along H)."""
    b = _a(boxes_2d, dtype)
    v1, u1, v2, u2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    roi_h, roi_w = roi_size
    hu, hv = (u2 - u1) / dtype(roi_w) / dtype(2), (v2 - v1) / dtype(roi_h) / dtype(2)
    gu = tf_linspace(u1 + hu, u2 - hu, roi_w, dtype)  # reference passes roi_size[0]; maps are square
    gv = tf_linspace(v1 + hv, v2 - hv, roi_h, dtype)
    n = b.shape[0]
    out = np.empty((n, roi_h, roi_w, 2), dtype)
    out[..., 0] = gu[:, None, :]
    out[..., 1] = gv[:, :, None]
    return out


def proj_err_maps_norm(xyz_global, boxes_2d, cam_p, valid_mask, dtype=np.float32):
    """monopsr_output_builder.py:681-746 get_proj_err_maps_norm -> (proj_err_norm (N,), proj_err_maps_norm (N,H,W,2)).
    expected uv - projected uv, divided by the box [w, h], masked, clipped to [-2, 2], summed per instance and
    divided by the number of valid pixels (at least 1)."""
    x = _a(xyz_global, dtype)
    n, h, w, _ = x.shape
    b = _a(boxes_2d, dtype)
    m = _a(valid_mask, dtype).reshape(n, h, w, 1)
    exp_uv = tf_get_exp_proj_uv_map(b, (h, w), dtype)
    uv = project_pc_to_image(x.reshape(n, h * w, 3).transpose(0, 2, 1), cam_p, dtype)
    uv = uv.transpose(0, 2, 1).reshape(n, h, w, 2)
    err = exp_uv - uv
    wh = np.stack([b[:, 3] - b[:, 1], b[:, 2] - b[:, 0]], 1).reshape(n, 1, 1, 2)
    err = np.clip(err / wh * m, dtype(-2), dtype(2))
    nv = m.sum((1, 2, 3), dtype=dtype)
    nv = np.where(nv < 1, dtype(1), nv)
    return (err.sum((1, 2, 3), dtype=dtype) / nv).astype(dtype), err.astype(dtype)


def inst_depth_map_local_to_global(depth_local, global_depth, boxes_2d, view_angs, cam_p, rotate_view=True,
                                   dtype=np.float32):
    """instance_utils.py:605-680 tf_inst_depth_map_local_to_global.  (N,H,W,1) local depth + centroid depth + (when
    rotate_view) the view-normalisation offset, which the reference interpolates between the box's left and right
    edge rays with H samples and then lays out along the map's ROW axis (reshape to (N,H,1,1), tile over columns);
    that layout is kept as is."""
    d = _a(depth_local, dtype)
    n, h, w = d.shape[:3]
    d = d.reshape(n, h, w, 1)
    z = _a(global_depth, dtype).reshape(n, 1)
    if not rotate_view:
        return d + z.reshape(n, 1, 1, 1)
    cam_p = _a(cam_p, dtype).reshape(3, 4)
    cu, f = cam_p[0, 2], cam_p[0, 0]
    b = _a(boxes_2d, dtype)
    va = _a(view_angs, dtype).reshape(n, 1)
    x1, x2 = b[:, 1], b[:, 3]
    gs = (x2 - x1) / dtype(h) / dtype(2)
    x1, x2 = x1 + gs, x2 - gs
    val = np.arctan2((x1 - cu) / f, dtype(1)).reshape(n, 1)
    var = np.arctan2((x2 - cu) / f, dtype(1)).reshape(n, 1)
    xz = z / np.cos(va)
    off = []
    for v in (val, var):
        o = xz / np.cos(v - va)
        off.append((o * np.sin(v - va) * np.sin(va)).reshape(n))
    rows = tf_linspace(-off[0], -off[1], h, dtype)
    return (d + z.reshape(n, 1, 1, 1) + rows.reshape(n, h, 1, 1)).astype(dtype)


# ------------------------------------------------------------------------------------------ post-processing

def angle_bin_to_orientation(angle_bin, residual, num_bins):
    """orientation_encoder.py:83-107 np_angle_bin_to_orientation."""
    two_pi = 2 * np.pi
    angle = angle_bin * (two_pi / num_bins) + residual
    if angle < -np.pi:
        angle = angle + two_pi
    if angle > np.pi:
        angle = angle - two_pi
    return angle


def compute_box_3d_corners(box_3d):
    """obj_utils.py:835-864 (and :623-654): (3,8) corners of [x,y,z,l,w,h,ry]; y is the BOTTOM face centre."""
    tx, ty, tz, l, w, h, ry = [float(v) for v in box_3d]
    rot = np.array([[np.cos(ry), 0, np.sin(ry)], [0, 1, 0], [-np.sin(ry), 0, np.cos(ry)]])
    xc = np.array([l, l, -l, -l, l, l, -l, -l]) / 2
    yc = np.array([0, 0, 0, 0, -h, -h, -h, -h])
    zc = np.array([w, -w, -w, w, w, -w, -w, w]) / 2
    return rot @ np.array([xc, yc, zc]) + np.array([[tx], [ty], [tz]])


def project_to_image_space(box_3d, cam_p, image_size):
    """box_3d_projector.py:14-98 with truncate=True, discard=True, discard_before_truncation=True.
    image_size = (w, h).  Returns [x1,y1,x2,y2] or None."""
    uv = project_pc_to_image(compute_box_3d_corners(box_3d), cam_p)
    box = np.array([uv[0].min(), uv[1].min(), uv[0].max(), uv[1].max()])
    iw, ih = image_size
    if box[0] > iw or box[1] > ih or box[2] < 0 or box[3] < 0:
        return None
    if box[2] - box[0] > iw * 0.8 or box[3] - box[1] > ih * 0.8:
        return None
    box[0], box[1] = max(box[0], 0), max(box[1], 0)
    box[2], box[3] = min(box[2], iw), min(box[3], ih)
    return box


def postprocess_cen_x(box_2d, box_3d, cam_p):
    """instance_utils.py:988-1032: centroid u from the ratio at which the projected centroid splits the projected
    3-D box's width, applied to the detected 2-D box, back-projected at the predicted depth."""
    cam_p = np.asarray(cam_p, np.float64).reshape(3, 4)
    f, cu = cam_p[0, 0], cam_p[0, 2]
    cu_uv = project_pc_to_image(compute_box_3d_corners(box_3d), cam_p)
    cen_uv = project_pc_to_image(np.asarray(box_3d[0:3], np.float64).reshape(3, 1), cam_p)
    umin, umax = cu_uv[0].min(), cu_uv[0].max()
    ratio = (cen_uv[0, 0] - umin) / (umax - umin)
    u = box_2d[1] + ratio * (box_2d[3] - box_2d[1])
    return (u - cu) * (box_3d[2] / f)


def score_boxes(img_shape, boxes_2d, boxes_3d, valid_scores, cam_p, max_depth=45.0):
    """monopsr_output_builder.py:805-860 (the frame's P2 is passed in instead of read from the dataset)."""
    out = np.zeros(len(boxes_2d), np.float64)
    for i, (b2, b3) in enumerate(zip(boxes_2d, boxes_3d)):
        pb = project_to_image_space(b3, cam_p, (img_shape[1], img_shape[0]))
        iou = np.asarray(b2, np.float64)[[1, 0, 3, 2]]
        if pb is None:
            fit = 0.1
        else:
            hh, ww = iou[3] - iou[1], iou[2] - iou[0]
            fit = 1.0 - (abs((iou[0] - pb[0]) / ww) + abs((iou[2] - pb[2]) / ww)
                         + abs((iou[1] - pb[1]) / hh) + abs((iou[3] - pb[3]) / hh))
        sd = np.clip(1.0 - b3[2] / max_depth, 0.1, 1.0)
        out[i] = 0.95 * float(np.ravel(valid_scores)[i]) + 0.05 * (sd + fit) / 2.0
    return out


def format_predictions(lwh, view_angs, alpha_bins, alpha_regs, centroids, boxes_2d, scores, class_indices, cam_p,
                       img_shape, num_alpha_bins=12, centroid_type="middle", post_process_cen_x=True):
    """monopsr_model.py:960-1071 format_predictions in 'test' mode for the outputs of model 000
    (lwh offset, alpha 'dc', view_ang 'est', centroids xyz): -> (box_3d (N,9), box_2d (N,7)) where
    box_3d = [x,y,z,l,w,h,ry,score,class-1] and box_2d = [y1,x1,y2,x2,alpha,score,class-1]."""
    n = len(boxes_2d)
    b3 = np.zeros((n, 7), np.float64)
    b3[:, 3:6] = lwh
    best = np.argmax(alpha_bins, 1)
    alphas = np.array([angle_bin_to_orientation(int(k), float(alpha_regs[i, k]), num_alpha_bins)
                       for i, k in enumerate(best)])
    b3[:, 6] = alphas + np.ravel(view_angs)
    cen = np.array(centroids, np.float64)
    if centroid_type == "middle":
        cen[:, 1] += b3[:, 5] / 2
    b3[:, 0:3] = cen
    if post_process_cen_x:
        b3[:, 0] = [postprocess_cen_x(b2, bb, cam_p) for b2, bb in zip(np.asarray(boxes_2d, np.float64), b3)]
    sc = score_boxes(img_shape, boxes_2d, b3, scores, cam_p).reshape(n, 1)
    cls = np.asarray(class_indices, np.float64).reshape(n, 1) - 1
    return (np.hstack([b3, sc, cls]),
            np.hstack([np.asarray(boxes_2d, np.float64), alphas.reshape(n, 1), sc, cls]))


# ------------------------------------------------------------------------------------- orientation encoding

def wrap_to_pi(angles):
    """orientation_encoder.py:6-8 np_wrap_to_pi."""
    return (np.asarray(angles) + np.pi) % (2 * np.pi) - np.pi


def orientation_to_angle_bin(orientation, num_bins, overlap=0.0):
    """orientation_encoder.py:11-81 np_orientation_to_angle_bin -> (best bin, residual to EVERY bin centre, one-hot
    of the valid bins).  Bin 0 is centred on angle 0.  With overlap, the upper neighbour is added whenever the
    angle is within `overlap` of the upper boundary; the lower neighbour only when it wraps to the last bin (the
    reference appends inside the wrap branch, :69-73) -- kept as is."""
    two_pi = 2 * np.pi
    wrapped = orientation % two_pi
    per_bin = two_pi / num_bins
    shifted = (wrapped + per_bin / 2) % two_pi
    best = int(shifted / per_bin)
    best_res = shifted - (best * per_bin + per_bin / 2)
    centres = np.asarray([per_bin * k for k in range(num_bins)])
    residuals = np.arctan2(np.sin(wrapped - centres), np.cos(wrapped - centres))
    valid = [best]
    if overlap != 0.0:
        centre = best * per_bin
        actual = best * per_bin + best_res
        if abs(centre + 0.5 * per_bin - actual) < overlap:
            valid.append(0 if best + 1 == num_bins else best + 1)
        elif abs(centre - 0.5 * per_bin - actual) < overlap:
            if best - 1 < 0:
                valid.append(num_bins - 1)
    one_hot = np.zeros(num_bins)
    one_hot[np.asarray(valid)] = 1
    return best, residuals, one_hot
