"""KITTI label-file export of the detections `MonoPSRModel.format_predictions` produces.

Replaces `save_predictions_box_3d_in_kitti_format` of the reference (core/evaluator_utils.py:114-277): same call
signature, same output tree (`<base>/kitti_predictions_3d/<split>/<threshold>/<step>/data/<sample>.txt`) and the same
file text (16 columns, values rounded to 3 decimals, truncation / occlusion -1, CRLF line ends, empty file for a frame
without detections).  Built differently: the conversion is a pure function over the in-memory (n,9) / (n,7) arrays
(`kitti_label_rows`), so predictions can be exported straight from `format_predictions` output with
`export_kitti_labels` -- no detour through the '%0.5f' text files -- and the file-based entry point is a thin reader
in front of it.  `project_3d_box=True` (2-D box = clipped projection of the 3-D box, box_3d_projector.py:14-95) is
vectorised over the boxes of a frame; it needs the frame's P2 matrix and image size, which the caller supplies
(`frame_info`) because this package has no dataset reader.

Running the native KITTI evaluator and the metrics CSV (evaluator_utils.py:280-560) are out of scope.
"""
import os

import numpy as np

# box_3d row: x y z l w h ry score class   |   box_2d row: y1 x1 y2 x2 alpha score class
_X, _L, _W, _H, _RY, _SCORE, _CLS = 0, 3, 4, 5, 6, 7, 8


def project_boxes_3d(boxes_3d, cam_p, image_size, max_fraction=0.8):
    """Image-space bounding boxes [x1, y1, x2, y2] of 3-D boxes (n,7+) [x, y, z, l, w, h, ry] (y = bottom face),
    clipped to the image; `keep` marks the boxes the reference would keep: at least partly inside the image and,
    before clipping, not wider / taller than `max_fraction` of it, with a clipped extent that is not degenerate.
    Returns (boxes (n,4), keep (n,) bool)."""
    b = np.asarray(boxes_3d, np.float64).reshape(-1, boxes_3d.shape[-1])
    n = b.shape[0]
    sx = np.array([1, 1, -1, -1, 1, 1, -1, -1]) * 0.5
    sz = np.array([1, -1, -1, 1, 1, -1, -1, 1]) * 0.5
    sy = np.array([0, 0, 0, 0, -1, -1, -1, -1.0])
    cx, cy, cz = b[:, _L, None] * sx, b[:, _H, None] * sy, b[:, _W, None] * sz  # (n,8) box frame
    c, s = np.cos(b[:, _RY, None]), np.sin(b[:, _RY, None])
    pts = np.stack([c * cx + s * cz + b[:, 0, None], cy + b[:, 1, None], -s * cx + c * cz + b[:, 2, None],
                    np.ones((n, 8))], axis=1)  # (n,4,8) camera frame, homogeneous
    uvw = np.einsum('ij,njk->nik', np.asarray(cam_p, np.float64).reshape(3, 4), pts)
    u, v = uvw[:, 0] / uvw[:, 2], uvw[:, 1] / uvw[:, 2]
    raw = np.stack([u.min(1), v.min(1), u.max(1), v.max(1)], axis=1)
    w_img, h_img = float(image_size[0]), float(image_size[1])
    inside = (raw[:, 0] <= w_img) & (raw[:, 1] <= h_img) & (raw[:, 2] >= 0) & (raw[:, 3] >= 0)
    small = ((raw[:, 2] - raw[:, 0]) <= max_fraction * w_img) & ((raw[:, 3] - raw[:, 1]) <= max_fraction * h_img)
    clipped = np.stack([np.clip(raw[:, 0], 0, None), np.clip(raw[:, 1], 0, None),
                        np.clip(raw[:, 2], None, w_img), np.clip(raw[:, 3], None, h_img)], axis=1)
    return clipped, inside & small


def kitti_label_rows(box_3d, box_2d, classes, score_threshold, image_boxes=None, keep=None):
    """Detections of one frame -> list of KITTI label lines (no line ends).
    box_3d (n,9), box_2d (n,7) as format_predictions returns them; `image_boxes` (n,4) [x1,y1,x2,y2] replaces the
    2-D boxes of box_2d (projection mode), `keep` (n,) drops rows on top of the score filter."""
    box_3d = np.asarray(box_3d, np.float64).reshape(-1, 9)
    box_2d = np.asarray(box_2d, np.float64).reshape(-1, 7)
    sel = box_3d[:, _SCORE] >= score_threshold
    if keep is not None:
        sel &= np.asarray(keep, bool)
    b3, b2 = box_3d[sel], box_2d[sel]
    xyxy = b2[:, [1, 0, 3, 2]] if image_boxes is None else np.asarray(image_boxes, np.float64)[sel]
    # alpha | x1 y1 x2 y2 | h w l | x y z | ry score
    numbers = np.round(np.column_stack([b2[:, 4], xyxy, b3[:, [_H, _W, _L]], b3[:, _X:_X + 3],
                                        b3[:, [_RY, _SCORE]]]), 3)
    names = [classes[int(k)] for k in b3[:, _CLS]]
    return [' '.join([name, '-1', '-1'] + [repr(float(v)) for v in row]) for name, row in zip(names, numbers)]


def write_kitti_label_file(path, rows):
    with open(path, 'w', newline='') as f:
        f.write(''.join(r + '\r\n' for r in rows))


def kitti_output_dir(predictions_base_dir, data_split, score_threshold, global_step):
    return '{}/kitti_predictions_3d/{}/{}/{}/data'.format(predictions_base_dir, data_split,
                                                          round(score_threshold, 3), global_step)


def export_kitti_labels(predictions, classes, score_threshold, out_dir, sample_names=None, project_3d_box=False,
                        frame_info=None):
    """predictions: {sample name: (box_3d (n,9), box_2d (n,7))} straight from format_predictions.  One file per name
    in `sample_names` (default: the dict's keys); frames without surviving detections get an empty file.
    project_3d_box: frame_info(sample name) -> (cam_p (3,4), (image_w, image_h)).  Returns the number of frames
    with at least one detection."""
    if project_3d_box and frame_info is None:
        raise ValueError('project_3d_box=True needs frame_info(sample_name) -> (cam_p, (image_w, image_h))')
    score_threshold = round(score_threshold, 3)
    os.makedirs(out_dir, exist_ok=True)
    valid = 0
    for name in (list(predictions) if sample_names is None else sample_names):
        rows = []
        if name in predictions and len(predictions[name][0]):
            b3, b2 = predictions[name]
            boxes, keep = (None, None)
            if project_3d_box:
                cam_p, size = frame_info(name)
                boxes, keep = project_boxes_3d(np.asarray(b3, np.float64).reshape(-1, 9), cam_p, size)
            rows = kitti_label_rows(b3, b2, classes, score_threshold, boxes, keep)
        valid += bool(rows)
        write_kitti_label_file(os.path.join(out_dir, name + '.txt'), rows)
    return valid


def _read_rows(path, width):
    if not os.path.exists(path) or os.path.getsize(path) == 0:
        return np.zeros((0, width))
    return np.loadtxt(path, ndmin=2).reshape(-1, width)


def save_predictions_box_3d_in_kitti_format(score_threshold, dataset, predictions_base_dir, predictions_box_3d_dir,
                                            predictions_box_2d_dir, global_step, project_3d_box=False):
    """The reference's entry point (same arguments): converts the per-sample text files MonoPSRModel.save_predictions
    wrote.  dataset: any object with `data_split`, `num_samples`, `sample_list[i].name`, `classes`; for
    project_3d_box=True also `frame_info(sample_name) -> (cam_p, (image_w, image_h))`.  Returns the output directory."""
    names = [dataset.sample_list[i].name for i in range(dataset.num_samples)]
    predictions = {}
    for name in names:
        b3 = _read_rows(os.path.join(predictions_box_3d_dir, name + '.txt'), 9)
        if len(b3):
            predictions[name] = (b3, _read_rows(os.path.join(predictions_box_2d_dir, name + '.txt'), 7))
    out_dir = kitti_output_dir(predictions_base_dir, dataset.data_split, score_threshold, global_step)
    export_kitti_labels(predictions, dataset.classes, score_threshold, out_dir, names, project_3d_box,
                        getattr(dataset, 'frame_info', None))
    return out_dir
