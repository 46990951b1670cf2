"""Training with BOTH trunks (the reference's whole parameter set): crop_and_resize image gradient, and a training step
from a raw image + boxes whose gradients reach the full-image trunk."""
import numpy as np
import pytest
import torch

from oracle import net as onet

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("shape,crop", [((1, 10, 38, 8), (6, 6)), ((2, 7, 9, 3), (5, 4)), ((1, 12, 12, 4), (1, 1))])
def test_crop_and_resize_image_gradient(shape, crop):
    """mpsr_crop_and_resize_grad vs float64 autograd through the CPU restatement, incl. boxes leaving the image."""
    from monopsr_amd.core import autograd_ops as ops
    rng = np.random.default_rng(shape[1])
    img = rng.standard_normal(shape).astype(np.float32)
    nb = 7
    y1, x1 = rng.uniform(-0.2, 0.8, nb), rng.uniform(-0.2, 0.8, nb)
    boxes = np.stack([y1, x1, y1 + rng.uniform(0.05, 0.6, nb), x1 + rng.uniform(0.05, 0.6, nb)], 1).astype(np.float32)
    ind = rng.integers(0, shape[0], nb).astype(np.int32)
    up = rng.standard_normal((nb, crop[0], crop[1], shape[3])).astype(np.float32)
    x64 = torch.from_numpy(img).double().requires_grad_()
    ref = onet.tf_crop_and_resize(x64, boxes, ind, crop[0], crop[1], 0.0)
    (ref * torch.from_numpy(up).double()).sum().backward()
    xt = _dev(img).requires_grad_()
    got = ops.crop_and_resize(xt, _dev(boxes), _dev(ind), crop, 0.0)
    np.testing.assert_allclose(got.detach().cpu().numpy(), ref.detach().numpy(), atol=1e-5)
    (got * _dev(up)).sum().backward()
    want = x64.grad.numpy()
    assert np.abs(xt.grad.cpu().numpy() - want).max() < 1e-5 * max(1.0, np.abs(want).max())


def test_training_step_with_both_trunks():
    """Image + boxes in (no precomputed feature crop): forward through both ResNets, loss, backward; every layer of
    BOTH trunks receives a gradient, the flat gradient buffer covers the whole parameter set, and a step lowers the
    loss."""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    B, div = 3, 8
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=111, width_div=div, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    net = train_net.TrainNet(weights, width_div=div, full_trunk=True)
    crop_only = train_net.TrainNet(weights, width_div=div)
    assert net.params.numel() > crop_only.params.numel() and net.full_base == len(crop_only.layers)
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, lr=1e-4)
    rng = np.random.default_rng(112)
    H, Wd = 375, 1242
    y1, x1 = rng.uniform(100, 200, B), rng.uniform(100, 900, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(40, 120, B), x1 + rng.uniform(60, 200, B)], 1).astype(np.float32)
    sample = dict(rgb_image=_dev(rng.integers(0, 256, (H, Wd, 3)).astype(np.float32)),
                  boxes_2d=_dev(boxes), boxes_2d_norm=_dev(boxes / np.array([H, Wd, H, Wd], np.float32)),
                  cam_p=_dev(np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)),
                  est_view_angs=_dev(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
                  class_indices=torch.ones((B, 1), dtype=torch.int32, device="cuda"),
                  mean_lwh=_dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                  prop_cen_z_offset=torch.full((B,), 2.178, device="cuda"))
    sample.update(trainer.synthetic_ground_truth(sample, seed=113))
    net.zero_grad()
    _, total = tr.loss(tr.forward(sample), sample)
    total.backward()
    tr.reducer.finish()
    empty = [i for i, L in enumerate(net.layers) if float(L.dw.abs().max()) == 0.0]
    assert not empty, empty
    # directional derivative: a small step against the gradient of ONE part of the buffer lowers the loss by
    # eps * |g| (checks sign and scale of the full-image trunk's gradient, crop_and_resize gradient included)
    g, p0, L0 = net.grads.clone(), net.params.clone(), float(total)
    lo = (net.layers[net.full_base].dw.data_ptr() - net.grads.data_ptr()) // 4
    for a, b in ((lo, g.numel()), (0, lo)):
        gs = torch.zeros_like(g)
        gs[a:b] = g[a:b]
        nrm = float(gs.double().norm())
        net.params.copy_(p0 - 1e-3 * gs / nrm)
        with torch.no_grad():
            L1 = float(tr.loss(tr.forward(sample), sample)[1])
        assert 0.9 < (L1 - L0) / (-1e-3 * nrm) < 1.1, ((a, b), L0, L1, nrm)
    net.params.copy_(p0)
    losses = [float(tr.step(sample)) for _ in range(6)]
    assert np.isfinite(losses).all() and min(losses[-3:]) < losses[0], losses


@pytest.mark.parametrize("shape,relu", [((3, 24, 24, 64), True), ((2, 7, 5, 8), False), ((1, 48, 48, 128), True)])
def test_batch_norm_training_mode_vs_float64_autograd(shape, relu):
    """conv -> BatchNorm(batch statistics, no scale, eps 1e-3) -> ReLU: outputs, input / weight / beta gradients and
    the moving-statistics update vs torch float64 (F.batch_norm, training=True)."""
    import torch.nn.functional as F
    from monopsr_amd.core import autograd_ops as ops
    rng = np.random.default_rng(shape[1])
    B, H, Wd, C = shape
    cin = 16
    x = (rng.standard_normal((B, H, Wd, cin)) * 3 + 5).astype(np.float32)     # off-centre: exercises the conditioning
    w = (rng.standard_normal((C, 9 * cin)) / np.sqrt(9 * cin)).astype(np.float32)
    beta = rng.standard_normal(C).astype(np.float32)
    mm, mv = rng.standard_normal(C).astype(np.float32), rng.uniform(0.5, 1.5, C).astype(np.float32)
    up = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    wt, bt = _dev(w), _dev(beta)
    dw, db = torch.zeros_like(wt), torch.zeros_like(bt)
    L = ops.LayerRef(wt, bt, dw, db, cin, C, 3, 3, 1, relu)
    L.batch_norm = ops.BatchNormState(_dev(mm), _dev(mv), 1e-3, 0.999)
    xt = _dev(x).requires_grad_()
    y = ops.conv2d(xt, L)
    (y * _dev(up)).sum().backward()
    # float64 reference
    x64 = torch.from_numpy(x).double().requires_grad_()
    w64 = torch.from_numpy(w).double().requires_grad_()
    b64 = torch.from_numpy(beta).double().requires_grad_()
    z = F.conv2d(x64.permute(0, 3, 1, 2), w64.reshape(C, 3, 3, cin).permute(0, 3, 1, 2), padding=1)
    rm, rv = torch.from_numpy(mm).double(), torch.from_numpy(mv).double()
    yr = F.batch_norm(z, rm, rv, None, b64, True, 1 - 0.999, 1e-3)
    if relu:
        yr = torch.relu(yr)
    yr = yr.permute(0, 2, 3, 1)
    (yr * torch.from_numpy(up).double()).sum().backward()
    rel = lambda a, b: float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
    assert rel(y.detach().cpu().numpy(), yr.detach().numpy()) < 2e-5
    assert rel(xt.grad.cpu().numpy(), x64.grad.numpy()) < 1e-4
    assert rel(dw.cpu().numpy(), w64.grad.numpy()) < 1e-4
    assert rel(db.cpu().numpy(), b64.grad.numpy()) < 1e-4
    assert rel(L.batch_norm.moving_mean.cpu().numpy(), rm.numpy()) < 1e-5          # F.batch_norm updated rm / rv in place
    assert rel(L.batch_norm.moving_variance.cpu().numpy(), rv.numpy()) < 1e-5


def test_training_step_with_decoder_batch_statistics():
    """TrainNet(decoder_bn='batch'): the reference's training graph for the map decoder.  Directional derivative of
    the whole loss along the decoder's gradient, beta included; moving statistics move."""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    B, div = 4, 4
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=121, width_div=div)
    net = train_net.TrainNet(weights, width_div=div, decoder_bn='batch')
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, lr=1e-4)
    rng = np.random.default_rng(122)
    y1, x1 = rng.uniform(100, 200, B), rng.uniform(100, 900, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(40, 120, B), x1 + rng.uniform(60, 200, B)], 1).astype(np.float32)
    sample = dict(rgb_image_crops=_dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)),
                  full_img_feature_crop=_dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0)
                                             .astype(np.float32)),
                  boxes_2d=_dev(boxes),
                  cam_p=_dev(np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)),
                  est_view_angs=_dev(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
                  class_indices=torch.ones((B, 1), dtype=torch.int32, device="cuda"),
                  mean_lwh=_dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                  prop_cen_z_offset=torch.full((B,), 2.178, device="cuda"))
    sample.update(trainer.synthetic_ground_truth(sample, seed=123))
    bn_layers = [L for L in net.layers if L.batch_norm is not None]
    assert len(bn_layers) == 4
    mm0 = bn_layers[0].batch_norm.moving_mean.clone()
    net.zero_grad()
    _, total = tr.loss(tr.forward(sample), sample)
    total.backward()
    tr.reducer.finish()
    assert not torch.equal(mm0, bn_layers[0].batch_norm.moving_mean)
    assert all(float(L.db.abs().max()) > 0 and float(L.dw.abs().max()) > 0 for L in bn_layers)
    g, p0, L0 = net.grads.clone(), net.params.clone(), float(total)
    first = bn_layers[0]
    last = net.layers[net.n_trunk + net.n_dec - 1]
    lo = (first.dw.data_ptr() - net.grads.data_ptr()) // 4
    hi = (last.dw.data_ptr() - net.grads.data_ptr()) // 4
    gs = torch.zeros_like(g)
    gs[lo:hi] = g[lo:hi]                       # the four BatchNorm layers: kernels and betas
    nrm = float(gs.double().norm())
    net.params.copy_(p0 - 1e-3 * gs / nrm)
    with torch.no_grad():
        L1 = float(tr.loss(tr.forward(sample), sample)[1])
    assert 0.9 < (L1 - L0) / (-1e-3 * nrm) < 1.1, (L0, L1, nrm)
    net.params.copy_(p0)
    losses = [float(tr.step(sample)) for _ in range(6)]
    assert np.isfinite(losses).all()


def test_export_weights_round_trips_into_the_inference_net(tmp_path):
    """TrainNet.export_weights() -> reference-named variables -> TensorFlow-format checkpoint -> DeviceNet computes
    what the TrainNet computes (frozen-BatchNorm form: the same function); the batch-statistics form exports its
    kernels, betas and moving statistics under the BatchNorm names."""
    from monopsr_amd.core import checkpoint_utils, train_net
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    div, B = 4, 3
    weights = W.synthetic_weights(seed=131, width_div=div, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    net = train_net.TrainNet(weights, width_div=div, full_trunk=True)
    g = torch.Generator(device="cuda").manual_seed(1)
    net.params.add_(torch.randn(net.params.shape, device="cuda", generator=g) * 1e-3 * net.params.abs().mean())
    exported = net.export_weights(width_div=div)
    assert set(exported) == set(weights)
    for k in weights:
        assert exported[k].shape == weights[k].shape and exported[k].dtype == np.float32, k
    prefix = checkpoint_utils.save_checkpoint(str(tmp_path / "monopsr"), exported, global_step=7)
    restored = W.synthetic_weights(seed=999, width_div=div, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    names = checkpoint_utils.restore_monopsr_weights(restored, checkpoint_utils.load_checkpoint(prefix))
    assert set(names) == set(weights)
    inf = dn.DeviceNet(restored, width_div=div, full_trunk=True)
    rng = np.random.default_rng(132)
    crops = _dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32))
    full = _dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0).astype(np.float32))
    img = _dev((rng.standard_normal((1, 160, 608, 3)) * 50).astype(np.float32))
    with torch.no_grad():
        a = net.trunk(crops)
        fa = net.squash_decoder(a, full)
        af = net.trunk(img, 'full')
    b = inf.trunk(crops)
    fb = inf.squash_decoder(b, full)
    bf = inf.trunk(img, 'full')
    rel = lambda x, y: float((x - y).abs().max() / (y.abs().max() + 1e-30))
    assert rel(b, a) < 1e-5 and rel(bf, af) < 1e-5
    for x, y in zip(fb, fa):
        assert rel(x, y) < 1e-5
    net_b = train_net.TrainNet(weights, width_div=div, decoder_bn='batch')
    eb = net_b.export_weights(width_div=div)
    n = "map_decoder/conv2/conv2_1"
    np.testing.assert_array_equal(eb[n + "/weights"], weights[n + "/weights"])
    np.testing.assert_array_equal(eb[n + "/BatchNorm/moving_variance"], weights[n + "/BatchNorm/moving_variance"])


def test_save_restore_step_keeps_the_schedule_and_the_moving_average():
    """Eight eager steps with an exponentially decaying learning rate and the parameter moving average, save, restore,
    one more step: the restore puts the average back IN PLACE (anyone holding the tensor keeps seeing it), the next
    step moves it by (1 - decay) x (params - average) like any other step, and a second save holds what the trainer
    holds.  (Until r05 this scenario was driven through a captured HIP graph of the step; the captured step was removed
    in r06 -- monopsr_amd/core/trainer.py: InstanceTrainer.step -- and its bounds, for the record, were: losses within
    4 x the spread of two eager runs + 2e-3, state within 4 x that spread + 0.01 / 0.03, at a learning rate of 2e-5
    because fixed bounds of 5e-3 (loss) and 0.03 / 0.10 (state) at 2e-4 failed one run in five for reasons of chaos.)"""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    B, div, steps = 4, 4, 8
    cfg = config_utils.default_config()
    opt = cfg.train_config.optimizer.adam_optimizer
    opt.learning_rate_type, opt.initial_learning_rate = 'exponential_decay', 2e-5
    opt.decay_steps, opt.decay_factor, opt.staircase = 2, 0.8, True
    opt.use_moving_average, opt.moving_average_decay = True, 0.9
    data = []
    for i in range(steps):
        rng = np.random.default_rng(500 + i)
        y1, x1 = rng.uniform(0, 150, B), rng.uniform(0, 1000, B)
        boxes = np.stack([y1, x1, y1 + rng.uniform(20, 200, B), x1 + rng.uniform(20, 200, B)], 1).astype(np.float32)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
        s = dict(rgb_image_crops=dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)),
                 full_img_feature_crop=dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0)
                                           .astype(np.float32)),
                 boxes_2d=dev(boxes), cam_p=dev(np.array([[721.5, 0, 609.6, 44.9], [0, 721.5, 172.9, 0.2],
                                                          [0, 0, 1, 0.003]], np.float32)),
                 est_view_angs=dev(rng.uniform(-0.6, 0.6, (B, 1)).astype(np.float32)),
                 class_indices=dev(np.ones((B, 1), np.int32)),
                 mean_lwh=dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                 prop_cen_z_offset=dev(np.full((B,), 2.178, np.float32)))
        s.update(trainer.synthetic_ground_truth(s, seed=600 + i))
        data.append(s)
    net = train_net.TrainNet(W.synthetic_weights(seed=77, width_div=div), width_div=div)
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config)
    losses = [float(tr.step(s)) for s in data]
    assert np.isfinite(losses).all() and tr.global_step == steps and net.step_count == steps
    assert abs(tr.optimizer.learning_rate(steps) - 2e-5 * 0.8 ** 4) < 1e-12  # staircase: 4 decays in 8 steps
    import tempfile
    with tempfile.TemporaryDirectory() as ckpt:
        prefix = tr.save(ckpt)
        shadow_ptr = tr.optimizer.shadow.data_ptr()
        before = tr.optimizer.shadow.clone()
        pb = net.params.clone()
        net.params.add_(1.0)  # (what a crash between save and resume leaves does not matter)
        tr.optimizer.shadow.zero_()
        assert tr.restore(prefix) == steps
        assert tr.optimizer.shadow.data_ptr() == shadow_ptr and torch.equal(tr.optimizer.shadow, before)
        assert torch.equal(net.params, pb) and net.step_count == steps
        float(tr.step(data[0]))
        torch.cuda.synchronize()
        assert bool(torch.isfinite(net.grads).all())
        want = before + (1.0 - 0.9) * (net.params - before)
        assert float((tr.optimizer.shadow - before).abs().max()) > 0, "the moving average froze after restore()"
        assert float((tr.optimizer.shadow - want).abs().max()) <= 1e-5 * float(want.abs().max())
        from monopsr_amd.core import tf_checkpoint
        saved = tf_checkpoint.read_checkpoint(tr.save(ckpt))
        np.testing.assert_array_equal(saved['monopsr_amd/flat_params/ExponentialMovingAverage'],
                                      tr.optimizer.shadow.cpu().numpy())
        # a checkpoint WITHOUT an average: the attribute is re-bound, the next step starts a new average
        tr.optimizer.shadow = None
        tr.restore(tr.save(ckpt))
        assert tr.optimizer.shadow is None
        float(tr.step(data[1]))
        assert tr.optimizer.shadow is not None


def test_weight_gradients_on_the_side_stream_are_complete_when_backward_returns():
    """r06: weight gradients run on a second HIP stream (autograd_ops._deposit_weight_grad).  The flat gradient buffer
    must be complete on the CALLER'S stream the moment backward() returns -- the end-of-pass callback joins the side
    stream -- so a reduction enqueued right behind backward() (no host synchronisation in between) sees every launch:
    its sum equals the sum taken after a device-wide synchronisation, and the gradients equal the ones computed with every
    launch on the main stream (up to the order of the kernels' atomics)."""
    from monopsr_amd.core import autograd_ops as ops
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    B, div = 8, 4
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=131, width_div=div)
    rng = np.random.default_rng(132)
    y1, x1 = rng.uniform(100, 200, B), rng.uniform(100, 900, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(40, 120, B), x1 + rng.uniform(60, 200, B)], 1).astype(np.float32)
    sample = dict(rgb_image_crops=_dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)),
                  full_img_feature_crop=_dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0).astype(np.float32)),
                  boxes_2d=_dev(boxes),
                  cam_p=_dev(np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)),
                  est_view_angs=_dev(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
                  class_indices=torch.ones((B, 1), dtype=torch.int32, device="cuda"),
                  mean_lwh=_dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                  prop_cen_z_offset=torch.full((B,), 2.178, device="cuda"))
    sample.update(trainer.synthetic_ground_truth(sample, seed=133))
    grads = {}
    assert ops.WGRAD_SIDE_STREAM
    try:
        for side in (True, False):
            ops.WGRAD_SIDE_STREAM = side
            net = train_net.TrainNet(weights, width_div=div, decoder_bn='batch')
            tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, lr=1e-4)
            for _ in range(3):  # (repeated: a missing join shows as a sum that changes after the synchronisation)
                net.zero_grad()
                _, total = tr.loss(tr.forward(sample), sample)
                total.backward()
                early = net.grads.double().abs().sum()  # enqueued on the caller's stream, no host sync before it
                assert not ops._JOIN_QUEUED
                torch.cuda.synchronize()
                late = net.grads.double().abs().sum()
                assert float(early) == float(late)
            grads[side] = net.grads.clone()
    finally:
        ops.WGRAD_SIDE_STREAM = True
    assert any(ops._WGRAD_STREAMS.values())
    scale = float(grads[False].abs().max())
    assert float((grads[True] - grads[False]).abs().max()) < 2e-5 * scale


def test_autograd_end_of_pass_callbacks_run_on_the_callers_stream():
    """What the weight-gradient stream's join relies on (autograd_ops._deposit_weight_grad): a callback queued from inside a
    backward formula runs at the end of the pass with the stream that was current where backward() was CALLED -- not the
    stream of the node that queued it -- so `current_stream().wait_stream(side)` there orders the caller behind the side
    stream.  A torch release that changes this must fail here, not as a race."""
    seen = {}

    class F(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen["node"] = torch.cuda.current_stream().cuda_stream
            torch.autograd.Variable._execution_engine.queue_callback(
                lambda: seen.__setitem__("callback", torch.cuda.current_stream().cuda_stream))
            return g * 2

    fwd, bwd = torch.cuda.Stream(), torch.cuda.Stream()
    x = torch.ones(4, device="cuda", requires_grad=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(fwd):
        y = F.apply(x).sum()
    bwd.wait_stream(fwd)
    with torch.cuda.stream(bwd):
        y.backward()
    torch.cuda.synchronize()
    assert seen["node"] == fwd.cuda_stream          # a node's backward runs on the stream of its forward
    assert seen["callback"] == bwd.cuda_stream      # the end-of-pass callback on the caller's


def test_side_streams_are_picked_by_measured_concurrency():
    """device_net.concurrent_stream: HIP maps streams onto a few hardware queues and two streams on one queue do not overlap
    at all (r06: with an RCCL communicator alive the weight-gradient stream landed on the main stream's queue and the step got
    SLOWER than with one stream).  The side streams of the package are therefore picked by a measurement -- two spin
    kernels behind a common event take one spin when they overlap, two when they are serialised -- and the pick must pass
    its own test, also with a dozen other streams alive (whatever queue they took)."""
    from monopsr_amd.core import device_net as dn
    dev = torch.device("cuda", 0)
    main = torch.cuda.current_stream(dev)
    others = [torch.cuda.Stream() for _ in range(12)]
    for s in others:  # make them real (a queue each, as far as the runtime has queues)
        with torch.cuda.stream(s):
            torch.zeros(1, device=dev)
    torch.cuda.synchronize()
    side = dn.concurrent_stream(dev)
    assert side.cuda_stream != main.cuda_stream
    assert dn.runs_concurrently(main, side)
    assert not dn.runs_concurrently(main, main)  # (the measurement itself: one stream is serialised with itself)
    # and relative to a non-default "main"
    other_main = others[3]
    side2 = dn.concurrent_stream(dev, other_main)
    assert dn.runs_concurrently(other_main, side2)
