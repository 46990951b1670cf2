"""BASELINE cfg4's arithmetic end to end (the r05 review lists cfg4 across ranks as untested): the WHOLE
training step of core/trainer.InstanceTrainer -- forward through both trunks, the reference's configured loss set, backward,
the bucketed gradient exchange of ReverseBucketReducer ("rccl" and "direct" modes) launched from inside backward, average,
per-variable clip AFTER the reduce (the reference's order, /root/reference/src/monopsr/core/trainer.py:76-81), Adam + moving
average -- on two ranks holding 8 instances each.

The invariant: a data-parallel step is the AVERAGE of the ranks' own reference steps.  (It is not "the unsharded step on
16 instances": the reference's map losses are means over valid pixels divided by the box count once more --
tf.losses.huber_loss(SUM_BY_NONZERO_WEIGHTS) / num_boxes, monopsr_model.py:554-958 -- so a term scales with 1 / B^2 and two
shards of 8 are not one batch of 16; every rank computing the reference's loss on ITS boxes is what sharding instances
means.)  So: (1) every rank's reduced gradient == the mean of the two gradients computed by single-process trainers on
shard 0 and shard 1 alone; (2) the parameters after the step == clip + Adam + moving average applied to that mean in one
process; (3) after a second step the ranks still hold identical bits.

One GPU: the two ranks compute on cuda:0 and meet over gloo (RCCL refuses two ranks on one device; what RCCL itself does
is covered on one rank by tests/test_rccl_one_rank_gpu.py).  Decoder BatchNorm on per-rank batch statistics (the
bench's default; pooled statistics have their own two-rank test, tests/test_bn_global_gpu.py), the image shared (a rank
owns boxes, not images).
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

B, DIV = 16, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _sample(full_trunk):
    from monopsr_amd.core import trainer
    rng = np.random.default_rng(41)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    H, Wd = 375, 1242
    y1, x1 = rng.uniform(100, 200, B), rng.uniform(100, 900, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(40, 120, B), x1 + rng.uniform(60, 200, B)], 1).astype(np.float32)
    s = dict(boxes_2d=dev(boxes),
             cam_p=dev(np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)),
             est_view_angs=dev(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
             class_indices=torch.ones((B, 1), dtype=torch.int32, device="cuda"),
             mean_lwh=dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
             prop_cen_z_offset=torch.full((B,), 2.178, device="cuda"))
    if full_trunk:
        s["rgb_image"] = dev(rng.integers(0, 256, (H, Wd, 3)).astype(np.float32))
        s["boxes_2d_norm"] = dev(boxes / np.array([H, Wd, H, Wd], np.float32))
    else:
        s["rgb_image_crops"] = dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32))
        s["full_img_feature_crop"] = dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // DIV)), 0).astype(np.float32))
    s.update(trainer.synthetic_ground_truth(s, seed=42))
    return s


def _make(full_trunk, mode):
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    cfg = config_utils.default_config()
    opt = cfg.train_config.optimizer.adam_optimizer
    opt.learning_rate_type, opt.initial_learning_rate = 'exponential_decay', 2e-5
    opt.decay_steps, opt.decay_factor, opt.staircase = 1, 0.5, True
    opt.use_moving_average, opt.moving_average_decay = True, 0.9
    scopes = (W.CROP_SCOPE, W.FULL_SCOPE) if full_trunk else (W.CROP_SCOPE,)
    net = train_net.TrainNet(W.synthetic_weights(seed=43, width_div=DIV, scopes=scopes), width_div=DIV,
                             full_trunk=full_trunk, decoder_bn='batch')
    # small buckets: the narrow net's flat buffer still spreads over several, launched from inside backward
    return trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config, clip_norm=1.0,
                                   bucket_bytes=1 << 20, allreduce=mode)


def _worker(rank, world, port, full_trunk, mode, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from monopsr_amd.core import data_parallel
        torch.cuda.set_device(0)
        tr = _make(full_trunk, mode)
        assert tr.reducer._active() and len(tr.reducer.buckets) > 3
        shard = data_parallel.shard_sample(_sample(full_trunk), rank, world)
        assert shard["boxes_2d"].shape[0] == B // world
        loss1 = float(tr.step(shard))
        grads1 = tr.net.grads.cpu().numpy().copy()  # (the fused update leaves the reduced gradient as it is)
        state1 = [t.cpu().numpy().copy() for t in (tr.net.params, tr.net.adam_m, tr.net.adam_v, tr.optimizer.shadow)]
        issued = list(tr.reducer.last_issued)
        loss2 = float(tr.step(shard))
        q.put((rank, loss1, grads1, state1, issued, loss2, tr.net.params.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def _rel(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize("full_trunk,mode", [(False, "rccl"), (True, "rccl"), (True, "direct")])
def test_two_rank_training_step_is_the_average_of_the_ranks_own_steps(full_trunk, mode):
    from monopsr_amd.core import data_parallel
    world, lr = 2, 2e-5
    # the ranks' own steps, one process each in turn (same initial weights: _make is seeded)
    shard_grads, shard_loss = [], []
    for r in range(world):
        tr = _make(full_trunk, mode)
        assert not tr.reducer._active()
        shard_loss.append(float(tr.step(data_parallel.shard_sample(_sample(full_trunk), r, world))))
        shard_grads.append(tr.net.grads.clone())
    mean_grad = (shard_grads[0] + shard_grads[1]) / 2
    # clip + Adam + moving average of that mean, in one process
    ref = _make(full_trunk, mode)
    ref.net.grads.copy_(mean_grad)
    ref.optimizer.apply_clipped_gradients(ref.net, 0, ref._clip_table(), ref.clip_norm)
    want_state = [t.cpu().numpy() for t in (ref.net.params, ref.net.adam_m, ref.net.adam_v, ref.optimizer.shadow)]
    mean_grad = mean_grad.cpu().numpy()

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, full_trunk, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = {}
    for _ in range(world):
        item = q.get(timeout=900)
        res[item[0]] = item[1:]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for r in range(world):
        loss1, grads1, state1, issued, loss2, params2 = res[r]
        assert abs(loss1 - shard_loss[r]) <= 1e-5 * abs(shard_loss[r]), (r, loss1, shard_loss[r])  # its own shard's loss
        assert _rel(grads1, mean_grad) < 2e-5, "reduced gradient of rank %d" % r  # (the atomics' summation order)
        for name, got, want in zip(("params", "adam_m", "adam_v", "average"), state1, want_state):
            d = np.abs(got - want)
            if name in ("params", "average"):
                # Adam's first step moves a parameter by ~lr * sign(g): where a gradient is at rounding level its sign
                # is free, so a few elements may differ by up to 2 lr; everything else to 1e-7
                assert d.max() <= 2.2 * lr and float((d > 1e-7).mean()) < 2e-3, (r, name, float(d.max()), float((d > 1e-7).mean()))
            else:
                assert _rel(got, want) < 1e-4, (r, name, _rel(got, want))
        assert np.isfinite(loss2)
        if mode == "rccl":
            assert set(issued) == {"all_reduce"} and len(issued) > 3
        else:  # 1 MiB buckets divide by two ranks: every bucket scatters and gathers (a last, odd one may all-reduce)
            assert issued.count("reduce_scatter_tensor") > 3
            assert issued.count("reduce_scatter_tensor") == issued.count("all_gather_into_tensor")
    # the ranks hold the same bits: after the exchange, after the update, and after a second step
    assert np.array_equal(res[0][1], res[1][1]), "the ranks disagree about the reduced gradient"
    for a, b in zip(res[0][2], res[1][2]):
        assert np.array_equal(a, b), "the ranks' state diverged in the first step"
    assert np.array_equal(res[0][5], res[1][5]), "the ranks' parameters diverged in the second step"
