"""Lists the host <-> device synchronisation points inside one call of a path (torch.cuda.set_sync_debug_mode): a
data-dependent index (`x[mask]`, `nonzero`), `.item()` or a `torch.tensor(list, device=...)` in a step stalls the launch
queue and leaves the GPU idle while the host catches up.

    python tools/sync_probe.py train | full_image
"""
import os
import sys
import traceback
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from monopsr_amd.core import config_utils, train_net, trainer  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402


def probe(fn):
    seen = {}

    def showwarning(message, category, filename, lineno, file=None, line=None):
        st = [f for f in traceback.extract_stack() if "/monopsr_amd/" in f.filename]
        key = tuple((f.filename.split("/monopsr_amd/")[-1], f.lineno) for f in st[-3:])
        seen[key] = seen.get(key, 0) + 1
    old = warnings.showwarning
    warnings.showwarning = showwarning
    warnings.simplefilter("always")
    torch.cuda.set_sync_debug_mode("warn")
    try:
        fn()
    finally:
        torch.cuda.set_sync_debug_mode("default")
        warnings.showwarning = old
    for k, v in sorted(seen.items(), key=lambda kv: -kv[1]):
        print(v, k)
    print("distinct synchronising sites:", len(seen))


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "train"
    dev = torch.device("cuda:0")
    cfg = config_utils.default_config()
    if what == "train":
        net = train_net.TrainNet(W.synthetic_weights(seed=0), device=dev)
        tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config)
        inp, _ = bench.make_inputs(64, 1024, 0, dev)
        sample = dict(rgb_image_crops=inp["crops"], full_img_feature_crop=inp["full_feat"], boxes_2d=inp["boxes"],
                      cam_p=inp["cam_p"], est_view_angs=inp["view"], class_indices=inp["cls"], mean_lwh=inp["mean_lwh"],
                      prop_cen_z_offset=inp["z_off"])
        sample.update(trainer.synthetic_ground_truth(sample, seed=7))
        for _ in range(2):
            tr.step(sample)
        torch.cuda.synchronize()
        probe(lambda: tr.step(sample))
    else:
        from monopsr_amd.core import device_net as dn
        from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel
        weights = W.synthetic_weights(seed=0, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
        net = dn.DeviceNet(weights, device=dev, full_trunk=True)
        model = MonoPSRModel(cfg.model_config, cfg.dataset_config, net, 'test')
        g = torch.Generator(device=dev).manual_seed(11)
        inp, _ = bench.make_inputs(32, 1024, 0, dev)
        sample = dict(rgb_image=torch.randint(0, 256, (375, 1242, 3), device=dev, generator=g).float(),
                      boxes_2d=inp["boxes"], boxes_2d_norm=inp["boxes"] / torch.tensor([375.0, 1242.0, 375.0, 1242.0], device=dev),
                      cam_p=inp["cam_p"], est_view_angs=inp["view"], class_indices=inp["cls"], mean_lwh=inp["mean_lwh"],
                      prop_cen_z_offset=inp["z_off"])
        for _ in range(2):
            model.build(sample)
        torch.cuda.synchronize()
        probe(lambda: model.build(sample))


if __name__ == "__main__":
    main()
