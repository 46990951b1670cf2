"""Data-parallel TRAINING step of the instance path (secondary measurement; bench.py is the headline metric):
forward + Chamfer / smooth-L1 losses + backward + gradient all-reduce (RCCL when launched with torchrun) + clip +
Adam, synthetic inputs of BASELINE config 4 (256 crops per GPU, 1024-pt... here the full 2304-pt map vs GT map).

    python tools/train_bench.py [--batch 256] [--steps 5]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_bench.py
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from monopsr_amd.core import config_utils  # noqa: E402
from monopsr_amd.core import train_net, trainer  # noqa: E402
from monopsr_amd.core import weights as W  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-clip", action="store_true")
    ap.add_argument("--unfused-update", action="store_true",
                    help="A/B: clip, Adam and the moving average as three separate passes (InstanceTrainer.fused_update)")
    ap.add_argument("--unfused-relu-grads", action="store_true",
                    help="A/B: the ReLU gradients inside a bottleneck unit as elementwise passes")
    ap.add_argument("--unlinked-units", action="store_true",
                    help="A/B: bottleneck units not chained (every unit runs its own ReLU-gradient pass)")
    ap.add_argument("--decoder-bn", default="batch", choices=["batch", "frozen"],
                    help="map-decoder BatchNorm: batch statistics as in the reference's training graph (default), or "
                         "frozen moving statistics folded into the convolutions")
    ap.add_argument("--full-image", action="store_true",
                    help="train BOTH trunks from a raw 375x1242 image + `--batch` boxes (the reference's step shape at "
                         "--batch 32) instead of the crop trunk over a precomputed full-image feature crop")
    ap.add_argument("--wgrad-winograd", type=int, default=1, choices=[0, 1],
                    help="A/B: 0 = every weight gradient on the direct kernel (mpsr_debug_set_wgrad_winograd), 1 = the "
                         "decoder's dense 3x3 layers in the F(4x4,3x3) domain and block3's atrous layers in F(3x3,3x3)")
    ap.add_argument("--dgrad-bank", type=int, default=1, choices=[0, 1],
                    help="A/B: 0 = the data-gradient layout of each layer packed inside backward (one launch per "
                         "layer), 1 = all layers by one launch per step (autograd_ops.DgradBank)")
    ap.add_argument("--math", default="fp32", choices=["fp32", "bf16x3"],
                    help="contraction arithmetic of the forward and data-gradient convolutions (wgrad stays fp32)")
    ap.add_argument("--trunk-streams-max-boxes", type=int, default=0,
                    help="A/B with --full-image: box count up to which the full-image trunk runs on its own stream "
                         "(TrainNet.side_stream_max_boxes, default 64)")
    ap.add_argument("--one-stream-trunks", action="store_true",
                    help="A/B with --full-image: both trunks on one stream (TrainNet.two_stream_trunks = False)")
    ap.add_argument("--bn-mask-from-y", action="store_true",
                    help="A/B: BatchNorm's backward reads y for the ReLU mask instead of rebuilding it from z")
    ap.add_argument("--main-high-priority", action="store_true",
                    help="A/B: the whole step on a high-priority stream (the weight-gradient stream stays normal)")
    ap.add_argument("--wgrad-priority", type=int, default=0, help="A/B: priority of the weight-gradient stream (-1 high)")
    ap.add_argument("--wgrad-min-steps", type=int, default=-1,
                    help="debug: 32-pixel steps a slice of a 1x1 weight gradient reduces at least (library default 12; 0 = off)")
    ap.add_argument("--wgrad-target", type=int, default=0,
                    help="debug: workgroups per weight-gradient launch the pixel slicing aims at (default 768)")
    ap.add_argument("--one-rank-rccl", action="store_true",
                    help="a one-rank RCCL communicator with the gradient exchange forced on (the N > 1 step's collectives "
                         "as identities: what issuing them costs on one GPU)")
    ap.add_argument("--no-exchange", action="store_true",
                    help="with --one-rank-rccl / N > 1: the reducer tracks its buckets but issues nothing (reducer.enabled = False)")
    ap.add_argument("--issue-on-main", action="store_true",
                    help="A/B with --one-rank-rccl: the caller's stream waits for the weight-gradient stream and issues the "
                         "collectives itself (ReverseBucketReducer.issue_on_wgrad_stream = False)")
    ap.add_argument("--wgrad-main-stream", action="store_true",
                    help="A/B: weight gradients on the main stream (autograd_ops.WGRAD_SIDE_STREAM = False)")
    ap.add_argument("--wgrad-streams", type=int, default=1, help="A/B: side streams the weight gradients rotate over")
    ap.add_argument("--wgrad-direct", action="store_true",
                    help="A/B: 1x1 weight gradients on the LDS-free kernel (mpsr_debug_set_wgrad_direct(1))")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or args.one_rank_rccl:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        sys.stdout.flush()
        out_fd = os.dup(1)  # (RCCL prints its version block to the C stdout: keep the JSON line's stdout apart)
        os.dup2(2, 1)
        sys.stdout = os.fdopen(out_fd, "w")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if args.issue_on_main:
        trainer.ReverseBucketReducer.issue_on_wgrad_stream = False
    cfg = config_utils.default_config()
    from monopsr_amd import _lib
    _lib.set_conv_math(args.math)
    _lib.lib().mpsr_debug_set_wgrad_winograd(args.wgrad_winograd)
    _lib.lib().mpsr_debug_set_wgrad_direct(1 if args.wgrad_direct else 0)
    if args.wgrad_min_steps >= 0:
        _lib.lib().mpsr_debug_set_wgrad_min_steps(args.wgrad_min_steps)
    if args.wgrad_target:
        _lib.lib().mpsr_debug_set_wgrad_target(args.wgrad_target)
    from monopsr_amd.core import autograd_ops
    autograd_ops.WGRAD_SIDE_STREAMS = args.wgrad_streams
    autograd_ops.WGRAD_STREAM_PRIORITY = args.wgrad_priority
    if args.wgrad_main_stream:
        autograd_ops.WGRAD_SIDE_STREAM = False
    if args.bn_mask_from_y:
        autograd_ops.BN_MASK_FROM_Z = False
    if args.one_stream_trunks:
        train_net.TrainNet.two_stream_trunks = False
    scopes = (W.CROP_SCOPE, W.FULL_SCOPE) if args.full_image else (W.CROP_SCOPE,)
    net = train_net.TrainNet(W.synthetic_weights(seed=0, scopes=scopes), device=dev, full_trunk=args.full_image,
                             decoder_bn=args.decoder_bn, dgrad_bank=bool(args.dgrad_bank))
    net.linked_units = not args.unlinked_units
    if args.trunk_streams_max_boxes and hasattr(net, "side_stream_max_boxes"):
        net.side_stream_max_boxes = args.trunk_streams_max_boxes
    if args.unfused_relu_grads:
        from monopsr_amd.core import autograd_ops
        autograd_ops.FUSED_RELU_GRADS = False
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config,
                                clip_norm=0.0 if args.no_clip else 1.0, force_collectives=args.one_rank_rccl)
    tr.fused_update = not args.unfused_update
    if args.no_exchange:
        tr.reducer.enabled = False
    inp, _ = bench.make_inputs(args.batch, 1024, rank, dev)
    B = args.batch
    sample = dict(rgb_image_crops=inp["crops"], full_img_feature_crop=inp["full_feat"], boxes_2d=inp["boxes"],
                  cam_p=inp["cam_p"], est_view_angs=inp["view"], class_indices=inp["cls"], mean_lwh=inp["mean_lwh"],
                  prop_cen_z_offset=inp["z_off"])
    if args.full_image:
        del sample["rgb_image_crops"], sample["full_img_feature_crop"]
        g = torch.Generator(device=dev).manual_seed(11 + rank)
        sample["rgb_image"] = torch.randint(0, 256, (375, 1242, 3), device=dev, generator=g).float()
        sample["boxes_2d_norm"] = sample["boxes_2d"] / torch.tensor([375.0, 1242.0, 375.0, 1242.0], device=dev)
    sample.update(trainer.synthetic_ground_truth(sample, seed=7 + rank))
    losses = []
    if args.main_high_priority:
        hp = torch.cuda.Stream(device=dev, priority=-1)
        hp.wait_stream(torch.cuda.current_stream(dev))
        torch.cuda.set_stream(hp)
    for _ in range(args.warmup):
        losses.append(float(tr.step(sample)))
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = 0.0
    for _ in range(args.steps):
        h0 = time.perf_counter()
        losses.append(tr.step(sample))
        host += time.perf_counter() - h0  # (the call returns when everything is enqueued: the host's share of a step)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "instance-crops/sec (train: fwd+bwd+allreduce+Adam)",
                          "value": round(B * world * args.steps / dt, 1), "unit": "crops/s", "n_gpus": world,
                          "host_enqueue_ms_per_step": round(1e3 * host / args.steps, 2),
                          "ms_per_step": round(1e3 * dt / args.steps, 2), "params": int(net.params.numel()),
                          "grad_bytes": int(net.grads.numel() * 4),
                          "loss_first_last": [round(float(losses[0]), 4), round(float(losses[-1]), 4)]}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
