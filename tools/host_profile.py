"""cProfile of the HOST side of training steps at a small batch (where the step is host-bound): top functions by
cumulative and by own time.  python tools/host_profile.py [--batch 8] [--steps 5]"""
import argparse
import cProfile
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from monopsr_amd.core import config_utils, train_net, trainer  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top", type=int, default=45)
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
    # (backward on the calling thread, so that the profile sees inside the backward formulas)
    torch.autograd.set_multithreading_enabled(False)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(args.steps):
        tr.step(sample)
    torch.cuda.synchronize()
    pr.disable()
    for key in ("tottime", "cumulative"):
        print("==== by %s (per %d steps)" % (key, args.steps))
        st = pstats.Stats(pr, stream=sys.stdout)
        st.sort_stats(key).print_stats(args.top)


if __name__ == "__main__":
    main()
