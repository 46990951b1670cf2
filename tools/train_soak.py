"""Soak: N training steps of tools/train_bench.py's trainer; prints allocated / reserved device memory and the loss every
`--every` steps (the side-stream weight gradients pin their operands with record_stream: reserved memory must settle after
the first steps, not grow), and checks that every loss is finite.
    python tools/train_soak.py [--steps 200] [--batch 256] [--full-image]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from monopsr_amd.core import config_utils, train_net, trainer  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--every", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--full-image", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = config_utils.default_config()
    scopes = (W.CROP_SCOPE, W.FULL_SCOPE) if args.full_image else (W.CROP_SCOPE,)
    net = train_net.TrainNet(W.synthetic_weights(seed=0, scopes=scopes), device=dev, full_trunk=args.full_image,
                             decoder_bn="batch")
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config, clip_norm=1.0)
    inp, _ = bench.make_inputs(args.batch, 1024, 0, dev)
    sample = dict(rgb_image_crops=inp["crops"], full_img_feature_crop=inp["full_feat"], boxes_2d=inp["boxes"],
                  cam_p=inp["cam_p"], est_view_angs=inp["view"], class_indices=inp["cls"], mean_lwh=inp["mean_lwh"],
                  prop_cen_z_offset=inp["z_off"])
    if args.full_image:
        del sample["rgb_image_crops"], sample["full_img_feature_crop"]
        g = torch.Generator(device=dev).manual_seed(11)
        sample["rgb_image"] = torch.randint(0, 256, (375, 1242, 3), device=dev, generator=g).float()
        sample["boxes_2d_norm"] = sample["boxes_2d"] / torch.tensor([375.0, 1242.0, 375.0, 1242.0], device=dev)
    sample.update(trainer.synthetic_ground_truth(sample, seed=7))
    reserved = []
    for k in range(args.steps):
        loss = float(tr.step(sample))
        assert loss == loss and abs(loss) < 1e30, (k, loss)
        if k % args.every == 0 or k == args.steps - 1:
            torch.cuda.synchronize()
            reserved.append(torch.cuda.memory_reserved(dev))
            print("step %4d  loss %12.4f  allocated %7.1f MiB  reserved %7.1f MiB" % (
                k, loss, torch.cuda.memory_allocated(dev) / 2 ** 20, reserved[-1] / 2 ** 20), flush=True)
    grew = reserved[-1] - reserved[len(reserved) // 2]
    print("reserved memory, second half of the run: %+.1f MiB" % (grew / 2 ** 20))
    assert grew <= 64 << 20, "reserved device memory keeps growing"


if __name__ == "__main__":
    main()
