"""Where do the small ATen launches of a training step come from?  One step of tools/train_bench.py's trainer under
torch.profiler with Python stacks; prints, for every ATen operator whose device time per call is small, the calls per
step by the innermost monopsr_amd source line that issued it.  (r06: the step has ~450 launches of 2-5 us in its
head / loss / optimizer sections.)  usage: python tools/aten_census.py [--batch 256] [--top 60]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402
from monopsr_amd.core import config_utils, train_net, trainer  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = config_utils.default_config()
    net = train_net.TrainNet(W.synthetic_weights(seed=0), device=dev, decoder_bn="batch")
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config, clip_norm=1.0)
    inp, _ = bench.make_inputs(args.batch, 1024, 0, dev)
    sample = dict(rgb_image_crops=inp["crops"], full_img_feature_crop=inp["full_feat"], boxes_2d=inp["boxes"],
                  cam_p=inp["cam_p"], est_view_angs=inp["view"], class_indices=inp["cls"], mean_lwh=inp["mean_lwh"],
                  prop_cen_z_offset=inp["z_off"])
    sample.update(trainer.synthetic_ground_truth(sample, seed=7))
    for _ in range(3):
        tr.step(sample)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        tr.step(sample)
        torch.cuda.synchronize()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    by_site = collections.Counter()
    by_op = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::"):
            continue
        # leaf ATen ops that launched something
        if any(c.name.startswith("aten::") for c in ev.cpu_children):
            continue
        if not ev.kernels:
            continue
        site = "?"
        for fr in ev.stack or []:
            if "monopsr_amd/" in fr and "site-packages" not in fr and "dist-packages" not in fr:
                site = fr[fr.index("monopsr_amd/"):].split(",")[0].strip()
                break
        by_site[(site, ev.name)] += len(ev.kernels)
        by_op[ev.name] += len(ev.kernels)
    print("launches by ATen operator:", dict(by_op.most_common(25)))
    print("total ATen launches in the step:", sum(by_op.values()))
    for (site, op), n in by_site.most_common(args.top):
        print("%4d  %-28s %s" % (n, op, site))


if __name__ == "__main__":
    main()
