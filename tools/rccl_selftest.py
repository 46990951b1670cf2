"""One-rank RCCL run of everything an N > 1 step would call (the GPU boxes available to the build have one GPU; r05
review, "What's missing" 1): init_process_group("nccl", world_size=1, device_id=cuda:0) exactly as bench.py does for
N > 1, then

  * ReverseBucketReducer in both modes ("rccl": all_reduce per bucket; "direct": reduce_scatter_tensor followed at once
    by all_gather_into_tensor -- the `dist.get_backend() == "nccl"` branch of core/trainer.py that gloo never takes) with
    force_active=True over the FULL both-trunk flat gradient (TrainNet(full_trunk=True): the 401 MB buffer BASELINE cfg4
    names), driven bucket by bucket through layer_ready() in backward order with async_op=True: on one rank every
    collective is an identity, so the buffer must come back bit for bit;
  * a real training step (image + boxes, both trunks) through InstanceTrainer(force_collectives=True) in both modes
    against the same step without collectives;
  * gather_instances / reduce_metric_sums (core/data_parallel.py) with force_collective=True;
  * the 401 MB all-reduce alone, timed (a one-rank number: it says what the call costs, nothing about xGMI).

Prints ONE JSON line (tests/test_rccl_one_rank_gpu.py asserts on it; profiles/r06_rccl_one_rank.json is a committed
copy).  Reference ordering kept: average -> per-variable clip -> Adam (/root/reference/src/monopsr/core/trainer.py:76-81)."""
import json
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def main():
    # RCCL prints a version block to the C stdout when the communicator is created (buffered: it would land behind the
    # JSON line): keep the original stdout for the line, send everything else to stderr (as bench.py does)
    sys.stdout.flush()
    out_stream = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    width_div = int(os.environ.get("MPSR_SELFTEST_WIDTH_DIV", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
           "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()),
           "device": torch.cuda.get_device_name(0), "torch": torch.__version__, "hip": torch.version.hip}

    from monopsr_amd.core import config_utils, data_parallel, train_net, trainer
    from monopsr_amd.core import weights as W
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=0, width_div=width_div, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    B = 8
    rng = np.random.default_rng(3)
    H, Wd = 375, 1242
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    y1, x1 = rng.uniform(100, 200, B), rng.uniform(100, 900, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(40, 120, B), x1 + rng.uniform(60, 200, B)], 1).astype(np.float32)
    sample = dict(rgb_image=t(rng.integers(0, 256, (H, Wd, 3)).astype(np.float32)), boxes_2d=t(boxes),
                  boxes_2d_norm=t(boxes / np.array([H, Wd, H, Wd], np.float32)),
                  cam_p=t(np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)),
                  est_view_angs=t(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
                  class_indices=torch.ones((B, 1), dtype=torch.int32, device=dev),
                  mean_lwh=t(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                  prop_cen_z_offset=torch.full((B,), 2.178, device=dev))
    sample.update(trainer.synthetic_ground_truth(sample, seed=4))

    net = train_net.TrainNet(weights, device=dev, width_div=width_div, full_trunk=True, decoder_bn='batch')
    out["params"] = int(net.params.numel())
    out["grad_bytes"] = int(net.grads.numel() * 4)
    out["reference_trainable_params"] = 100204832
    out["params_note"] = ("the flat buffer holds the reference's whole trainable set (both ResNet-101 trunks, decoder, "
                          "heads; SURVEY 8(d) cfg4: 100,204,832) in this package's parameterisation: BatchNorm folded "
                          "into one bias per channel, every tensor padded to 256 bytes, FC inputs padded to 4")
    p0 = net.params.clone()
    ref_grads = None
    for mode in ("none", "rccl", "direct"):
        net.params.copy_(p0)
        net.adam_m.zero_()
        net.adam_v.zero_()
        net.step_count = 0
        tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, lr=1e-5, bucket_bytes=64 << 20,
                                     allreduce="rccl" if mode == "none" else mode, force_collectives=(mode != "none"))
        rec = {}
        if mode != "none":
            rec["hardware_queues"] = tr.hardware_queues  # (the trainer probed which streams a waiting collective holds up)
            red = tr.reducer
            assert red._active(), "a forced reducer must be active on a one-rank group"
            # (1) the bare reducer over the whole flat buffer, driven like backward drives it
            g = torch.Generator(device=dev).manual_seed(5)
            pattern = torch.randn(net.grads.shape, device=dev, generator=g)
            net.grads.copy_(pattern)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for li in range(len(net.layers) - 1, -1, -1):
                red.layer_ready(li)
            red.finish(average=True)
            torch.cuda.synchronize()
            rec["reducer_ms"] = round(1e3 * (time.perf_counter() - t0), 2)
            rec["buckets"] = len(red.buckets)
            rec["issued"] = {k: red.last_issued.count(k) for k in sorted(set(red.last_issued))}
            rec["issue_order_head"] = red.last_issued[:4]
            rec["buffer_bit_identical"] = bool(torch.equal(net.grads.view(torch.int32), pattern.view(torch.int32)))
            del pattern
        # (2) a real step through it
        losses = [float(tr.step(sample))]
        torch.cuda.synchronize()
        rec["grads_finite"] = bool(torch.isfinite(net.grads).all())
        if mode != "none":
            rec["step_issued"] = {k: tr.reducer.last_issued.count(k) for k in sorted(set(tr.reducer.last_issued))}
        if ref_grads is None:  # (the first step starts from the same weights in every mode: gradients comparable)
            ref_grads = net.grads.clone()
        else:
            rec["grad_max_rel_diff_vs_no_collectives"] = float((net.grads - ref_grads).abs().max() /
                                                               ref_grads.abs().max())
        losses.append(float(tr.step(sample)))  # (and a second one on the updated weights)
        rec["loss"] = [round(v, 3) for v in losses]
        out["reducer_" + mode] = rec

    # (3) the metric path's two helpers
    x = torch.arange(24, dtype=torch.float32, device=dev).reshape(8, 3)
    gathered = data_parallel.gather_instances(x, 8, force_collective=True)
    sums = data_parallel.reduce_metric_sums([torch.tensor(1.5, device=dev), torch.tensor(2.0, device=dev)],
                                            force_collective=True)
    out["gather_instances_identity"] = bool(torch.equal(gathered, x))
    out["reduce_metric_sums"] = [float(v) for v in sums]

    # (4) the full gradient all-reduced alone
    dist.all_reduce(net.grads)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        dist.all_reduce(net.grads)
    torch.cuda.synchronize()
    out["allreduce_alone_ms_one_rank"] = round(1e3 * (time.perf_counter() - t0) / 5, 3)
    dist.barrier()
    dist.destroy_process_group()
    out_stream.write(json.dumps(out) + "\n")
    out_stream.flush()


if __name__ == "__main__":
    main()
