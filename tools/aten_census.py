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
    # which source line asked for each operator: a dispatch mode sees every ATen call of the step (forward, and the
    # backward's calls under the autograd engine's own thread are attributed to the backward formula's node instead)
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    calls = collections.Counter()

    class Census(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            site = "(autograd engine)"
            for fr in reversed(traceback.extract_stack()):
                if fr.filename.startswith(os.path.join(root, "monopsr_amd")):
                    site = "%s:%d %s" % (os.path.relpath(fr.filename, root), fr.lineno, fr.name)
                    break
            calls[(site, str(func))] += 1
            return func(*args, **(kwargs or {}))

    with Census():
        tr.step(sample)
        torch.cuda.synchronize()
    skip = ("aten.view", "aten._unsafe_view", "aten.t.", "aten.transpose", "aten.expand", "aten.slice", "aten.select",
            "aten.unsqueeze", "aten.squeeze", "aten.detach", "aten.alias", "aten.as_strided", "aten.permute",
            "aten.empty", "aten.reshape", "aten.unbind", "aten.split", "aten._local_scalar_dense", "aten.is_", "aten.sym_",
            "aten.lift_fresh", "aten.new_empty", "aten.narrow", "aten.unfold", "aten.stride", "aten.size")
    by_site = collections.Counter()
    for (site, op), n in calls.items():
        if not op.startswith(skip):
            by_site[site] += n
    print("ATen calls that launch, by source line (views and allocations left out): total", sum(by_site.values()))
    for site, n in by_site.most_common(args.top):
        ops = collections.Counter({op: k for (s_, op), k in calls.items() if s_ == site and not op.startswith(skip)})
        print("%4d  %-70s %s" % (n, site, ", ".join("%s x%d" % (o.replace("aten.", ""), k) for o, k in ops.most_common(6))))
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        tr.step(sample)
        torch.cuda.synchronize()
    by_op = collections.Counter()
    t_op = collections.Counter()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or any(c.name.startswith("aten::") for c in ev.cpu_children) or not ev.kernels:
            continue
        by_op[ev.name] += len(ev.kernels)
        t_op[ev.name] += sum(k.duration for k in ev.kernels)
    print("launches by ATen operator (device us):", {k: (n, round(t_op[k], 1)) for k, n in by_op.most_common(25)})
    print("total ATen launches in the step: %d, %.1f us on the device" % (sum(by_op.values()), sum(t_op.values())))


if __name__ == "__main__":
    main()
