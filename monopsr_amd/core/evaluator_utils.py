"""KITTI-format export of saved detections, mirroring core/evaluator_utils.py:114-277 of the reference
(save_predictions_box_3d_in_kitti_format).  Pure file conversion on the host: it reads the '%0.5f' text files
MonoPSRModel.save_predictions wrote (box_3d (n,9) and box_2d (n,7) per sample) and writes one KITTI label file per
sample.  Running the native KITTI evaluator and the metrics CSV (:280-..) are out of scope.
"""
import os

import numpy as np


def save_predictions_box_3d_in_kitti_format(score_threshold, dataset, predictions_base_dir, predictions_box_3d_dir,
                                            predictions_box_2d_dir, global_step, project_3d_box=False):
    """dataset: any object with `data_split`, `num_samples`, `sample_list[i].name` and `classes`.
    Returns the output directory.  project_3d_box=True (2-D boxes re-derived by projecting the 3-D boxes, which
    needs the frame's image size and calibration from the dataset reader) is not built; the reference's evaluator
    calls this with the default False."""
    if project_3d_box:
        raise NotImplementedError('project_3d_box=True needs the dataset reader (image size + calibration)')
    score_threshold = round(score_threshold, 3)
    kitti_predictions_3d_dir = predictions_base_dir + '/kitti_predictions_3d/{}/{}/{}/data'.format(
        dataset.data_split, score_threshold, global_step)
    os.makedirs(kitti_predictions_3d_dir, exist_ok=True)
    num_valid_samples = 0
    for sample_idx in range(dataset.num_samples):
        sample_name = dataset.sample_list[sample_idx].name
        prediction_file = sample_name + '.txt'
        kitti_path = kitti_predictions_3d_dir + '/' + prediction_file
        path_3d = predictions_box_3d_dir + '/' + prediction_file
        path_2d = predictions_box_2d_dir + '/' + prediction_file
        if not os.path.exists(path_3d):
            np.savetxt(kitti_path, [])
            continue
        all_predictions_3d = np.loadtxt(path_3d)
        if len(all_predictions_3d) == 0:
            np.savetxt(kitti_path, [])
            continue
        all_predictions_3d = all_predictions_3d.reshape(-1, 9)
        all_predictions_2d = np.loadtxt(path_2d).reshape(-1, 7)
        score_filter = all_predictions_3d[:, 7] >= score_threshold
        all_predictions_3d = all_predictions_3d[score_filter]
        all_predictions_2d = all_predictions_2d[score_filter]
        if len(all_predictions_3d) == 0:
            np.savetxt(kitti_path, [])
            continue
        boxes_2d = all_predictions_2d[:, [1, 0, 3, 2]]  # [y1,x1,y2,x2] -> [x1,y1,x2,y2]
        num_valid_samples += 1
        # columns: type trunc occ alpha x1 y1 x2 y2 h w l x y z ry score; [0:3] are filled below
        kitti_predictions = np.zeros([len(all_predictions_3d), 16])
        obj_types = [dataset.classes[class_idx] for class_idx in all_predictions_3d[:, 8].astype(np.int32)]
        kitti_predictions[:, 3] = all_predictions_2d[:, 4]
        kitti_predictions[:, 4:8] = boxes_2d
        kitti_predictions[:, 8] = all_predictions_3d[:, 5]
        kitti_predictions[:, 9] = all_predictions_3d[:, 4]
        kitti_predictions[:, 10] = all_predictions_3d[:, 3]
        kitti_predictions[:, 11:14] = all_predictions_3d[:, 0:3]
        kitti_predictions[:, 14:16] = all_predictions_3d[:, 6:8]
        kitti_predictions = np.round(kitti_predictions, 3)
        kitti_empty_1 = -1 * np.ones((len(kitti_predictions), 2), dtype=np.int32)
        kitti_text_3d = np.column_stack([obj_types, kitti_empty_1, kitti_predictions[:, 3:16]])
        np.savetxt(kitti_path, kitti_text_3d, newline='\r\n', fmt='%s')
    return kitti_predictions_3d_dir
