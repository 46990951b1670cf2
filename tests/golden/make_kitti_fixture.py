"""Generates tests/golden/kitti_cfg1.npz -- BASELINE config 1: 32 KITTI proposal crops (48x48) + 512-point GT clouds.

Run in the build container only (reads the reference's own test fixture, the mini KITTI tree under
/root/reference/src/monopsr/tests/datasets/Kitti/object; the GPU box never has it):

    python tests/golden/make_kitti_fixture.py

What is extracted (numpy + PIL only; SURVEY.md 8(d) cfg 1):
  * proposals = label_2 2-D boxes of `Car` objects of the train.txt + val.txt frames, oversampled to 32 with
    np.random.default_rng(0) as kitti_dataset.py:301-308 does;
  * per proposal: the frame index, box [y1,x1,y2,x2] in pixels, camera matrix P2, the 3-D box, the viewing angle;
  * the 48x48 proposal crops (float32) cut from the mean-subtracted, 320x1216-resized frames with the oracle's
    restatement of the reference's preprocessing + tf.image.crop_and_resize (img_preprocessor.py:12-35,
    monopsr_model.py:222-226), and ONE full RGB frame (uint8, the frame with most proposals) so that the whole
    image -> preprocess -> crop path can also run on real pixels;
  * GT clouds = velodyne points inside the label's 3-D box, moved to the camera-2 frame (calib), translated to the
    box centre ('middle' centroid) and rotated by minus the viewing angle (view normalisation of
    instance_utils.py:439-473: tr = R_y(-view) . T(-centroid)), resampled to 512 points with default_rng(0);
  * chamfer_ref = the REFERENCE's own monopsr.core.distance_metrics.calc_chamfer_dist (imported from
    /root/reference) between each GT cloud and a deterministic perturbed copy of it (`pred_clouds`).
The file holds data only.
"""
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, "/root/reference/src")
from monopsr.core import distance_metrics  # noqa: E402

ROOT = "/root/reference/src/monopsr/tests/datasets/Kitti/object"
HERE = os.path.dirname(os.path.abspath(__file__))


def read_calib(path):
    d = {}
    for line in open(path):
        if ":" in line:
            k, v = line.split(":", 1)
            d[k.strip()] = np.array([float(x) for x in v.split()], np.float64)
    p2 = d["P2"].reshape(3, 4)
    r0 = np.eye(4)
    r0[:3, :3] = d["R0_rect"].reshape(3, 3)
    tr = np.eye(4)
    tr[:3, :4] = d["Tr_velo_to_cam"].reshape(3, 4)
    return p2, r0, tr


def main():
    frames = []
    for split in ("train.txt", "val.txt"):
        frames += open(os.path.join(ROOT, split)).read().split()
    frames = [f[:6] for f in " ".join(frames).replace("000217000001", "000217 000001").split()]
    objs = []
    images = {}
    for f in frames:
        lab = os.path.join(ROOT, "training", "label_2", f + ".txt")
        if not os.path.exists(lab):
            continue
        p2, r0, tr = read_calib(os.path.join(ROOT, "training", "calib", f + ".txt"))
        velo = np.fromfile(os.path.join(ROOT, "training", "velodyne", f + ".bin"), np.float32).reshape(-1, 4)
        pts = (r0 @ tr @ np.c_[velo[:, :3], np.ones(len(velo))].T).T[:, :3]  # rectified camera-0 frame
        for line in open(lab):
            t = line.split()
            if t[0] != "Car":
                continue
            x1, y1, x2, y2 = [float(v) for v in t[4:8]]
            h, w, l = [float(v) for v in t[8:11]]
            cx, cy, cz = [float(v) for v in t[11:14]]
            ry = float(t[14])
            # points inside the 3-D box (box frame: x along length, y down from the bottom face, z along width)
            d = pts - np.array([cx, cy, cz])
            c, s = np.cos(ry), np.sin(ry)
            bx = c * d[:, 0] - s * d[:, 2]
            bz = s * d[:, 0] + c * d[:, 2]
            inside = (np.abs(bx) <= l / 2) & (np.abs(bz) <= w / 2) & (d[:, 1] <= 0) & (d[:, 1] >= -h)
            if inside.sum() < 20:
                continue
            x_offset = -p2[0, 3] / p2[0, 0]
            cen = np.array([cx - x_offset, cy - h / 2.0, cz])  # camera-2 frame, 'middle' centroid
            view = np.arctan2(cen[0], cen[2])
            q = pts[inside] - np.array([x_offset, 0, 0]) - cen
            cv, sv = np.cos(-view), np.sin(-view)
            local = np.stack([cv * q[:, 0] + sv * q[:, 2], q[:, 1], -sv * q[:, 0] + cv * q[:, 2]], 1)
            objs.append(dict(frame=f, box2d=[y1, x1, y2, x2], p2=p2, box3d=[cx, cy, cz, l, w, h, ry], view=view,
                             cloud=local.astype(np.float32)))
        if any(o["frame"] == f for o in objs):
            images[f] = np.asarray(Image.open(os.path.join(ROOT, "training", "image_2", f + ".png")).convert("RGB"))
    assert objs, "no usable Car objects"
    rng = np.random.default_rng(0)
    n = len(objs)
    idx = np.hstack([np.arange(n), rng.choice(n, max(0, 32 - n), replace=True)])[:32]
    sel = [objs[i] for i in idx]
    used = sorted({o["frame"] for o in sel})
    clouds = []
    for o in sel:
        c = o["cloud"]
        clouds.append(c[rng.choice(len(c), 512, replace=len(c) < 512)])
    gt = np.stack(clouds).astype(np.float32)
    pred = (gt[:, ::-1] * np.float32(1.05) + rng.normal(0, 0.05, gt.shape)).astype(np.float32)
    chamfer = np.array([distance_metrics.calc_chamfer_dist(pred[i].astype(np.float64), gt[i].astype(np.float64))
                        for i in range(32)])
    out = {
        "frames": np.array(used),
        "frame_index": np.array([used.index(o["frame"]) for o in sel], np.int32),
        "boxes_2d": np.array([o["box2d"] for o in sel], np.float32),
        "cam_p": np.array([o["p2"] for o in sel], np.float32),
        "boxes_3d": np.array([o["box3d"] for o in sel], np.float32),
        "view_angs": np.array([o["view"] for o in sel], np.float32),
        "gt_clouds": gt, "pred_clouds": pred, "chamfer_ref": chamfer,
    }
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    import torch
    from oracle import net as onet
    crops = np.zeros((32, 48, 48, 3), np.float32)
    for f in used:
        img = torch.from_numpy(images[f].astype(np.float32)).unsqueeze(0) - torch.tensor(onet.KITTI_CHANNEL_MEANS)
        pre = onet.tf_resize_bilinear(img, 320, 1216, False)
        H, W = images[f].shape[:2]
        rows = [i for i, o in enumerate(sel) if o["frame"] == f]
        norm = out["boxes_2d"][rows] / np.array([H, W, H, W], np.float32)  # kitti_dataset.py:450
        crops[rows] = onet.tf_crop_and_resize(pre, norm, np.zeros(len(rows), np.int32), 48, 48).numpy()
    out["rgb_crops"] = crops
    counts = {f: sum(o["frame"] == f for o in sel) for f in used}
    best = max(used, key=lambda f: counts[f])
    out["full_frame"] = images[best]
    out["full_frame_index"] = np.int32(used.index(best))
    np.savez_compressed(os.path.join(HERE, "kitti_cfg1.npz"), **out)
    print("objects", n, "frames used", used, "file MB",
          os.path.getsize(os.path.join(HERE, "kitti_cfg1.npz")) / 1e6)
    print("chamfer_ref[:4]", chamfer[:4])


if __name__ == "__main__":
    main()
