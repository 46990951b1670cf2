"""One data-parallel training step of the instance path: forward -> losses -> backward (gradients land in the flat
buffer) -> bucketed RCCL all-reduce overlapped with the rest of backward -> per-variable norm clip -> fused Adam.

What it mirrors of the reference's training setup (core/trainer.py:58-81, builders/optimizer_builder.py:24-122,
configs/monopsr_model_000.yaml:100-152): model.build() in 'train' mode -> model.loss() (the configured weighted sum
of per-output losses) -> Adam with the exponential-decay learning rate and the parameter moving average ->
per-variable clip_by_norm(1.0) as slim.learning.create_train_op(clip_gradient_norm=1.0) applies it.
Not mirrored: TF summaries and the checkpoint schedule.
"""
import os

import torch
import torch.distributed as dist

from monopsr_amd import _lib
from monopsr_amd.builders import optimizer_builder
from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel


class ReverseBucketReducer:
    """All-reduce of a flat gradient buffer in buckets, launched from the END of the buffer towards the start as
    layers report their gradients ready -- backward visits layers in roughly reverse buffer order (heads, decoder,
    trunk), so the first buckets are on the wire while the trunk is still back-propagating."""

    def __init__(self, flat, layer_spans, bucket_bytes=64 << 20, group=None, mode="rccl", force_active=False):
        """layer_spans[i]: the element range(s) of `flat` layer i's backward writes -- one (lo, hi) or a list of them
        (weight gradient AND bias gradient: a bucket holding any part of either must wait for the layer).
        mode: "rccl" = one all_reduce per bucket (the library's own algorithm choice); "direct" = reduce-scatter of the
        bucket into 1 / world shards followed by an all-gather (on the fully connected xGMI mesh every rank
        then sends each peer its shard ONCE per phase over that peer's own link, instead of 2 (world - 1) ring steps
        bound by one link).  Same sums up to the reduction order; selectable with bench.py --allreduce.
        force_active: issue the collectives even on a ONE-rank group (they are then identities) -- the one way to run
        the real RCCL sequence on a box with a single GPU (tests/test_rccl_one_rank_gpu.py, bench.py `rccl_one_rank`)."""
        if mode not in ("rccl", "direct"):
            raise ValueError("reducer mode must be 'rccl' or 'direct'")
        self.flat, self.group, self.bucket_bytes, self.mode = flat, group, bucket_bytes, mode
        self.force_active = bool(force_active)
        self.issued = []  # names of the collectives issued since the last reset(), in issue order (tests read it)
        self._shards = {}
        self.enabled = True  # False: buckets are tracked but nothing is exchanged (timing the step without it)
        n = max(1, bucket_bytes // flat.element_size())
        self.buckets = [(lo, min(lo + n, flat.numel())) for lo in range(0, flat.numel(), n)]
        # layers overlapping each bucket
        self.members = [set() for _ in self.buckets]
        self.layer_buckets = []
        for li, spans in enumerate(layer_spans):
            if spans and isinstance(spans[0], int):
                spans = [spans]
            ids = sorted({bi for lo, hi in spans for bi, (blo, bhi) in enumerate(self.buckets)
                          if lo < bhi and hi > blo})
            self.layer_buckets.append(ids)
            for bi in ids:
                self.members[bi].add(li)
        self.reset()

    def reset(self):
        self.pending = [set(m) for m in self.members]
        self.launched = [False] * len(self.buckets)
        self.works = []
        self.deferred = []
        self.issued = []

    def _active(self):
        return (self.enabled and dist.is_available() and dist.is_initialized()
                and (dist.get_world_size(self.group) > 1 or self.force_active))

    def layer_ready(self, li):
        for bi in self.layer_buckets[li]:
            self.pending[bi].discard(li)
            if not self.pending[bi] and not self.launched[bi]:
                self._launch(bi)

    # The weight gradients run on a side stream (autograd_ops._deposit_weight_grad), so the collective must be ordered
    # behind IT.  True: the collectives are issued with that stream current (it first waits for the caller's stream, which
    # may hold deposits of its own: BatchNorm's beta, the upsampled convolutions): the exchange starts when the bucket's
    # last weight gradient is done and the caller's stream -- the data-gradient chain -- is not stalled.  False: the
    # caller's stream waits for the side stream and issues the collective itself (A/B: tools/train_bench.py --issue-on-main).
    issue_on_wgrad_stream = True

    def _launch(self, bi):
        self.launched[bi] = True
        if self._active():
            restore = None
            if self.flat.is_cuda:
                from monopsr_amd.core import autograd_ops
                side = autograd_ops.wgrad_stream_if_any(self.flat.device) if self.issue_on_wgrad_stream else None
                if side is None:
                    autograd_ops.join_wgrad_stream(self.flat.device)
                else:
                    cur = torch.cuda.current_stream(self.flat.device)
                    autograd_ops.order_behind_pass_streams(side, self.flat.device)  # (the caller's stream is one of them)
                    side.wait_stream(cur)
                    torch.cuda.set_stream(side)
                    restore = cur
            try:
                self._issue(bi)
            finally:
                if restore is not None:
                    torch.cuda.set_stream(restore)

    def _issue(self, bi):
        lo, hi = self.buckets[bi]
        world = dist.get_world_size(self.group)
        if self.mode == "direct" and (hi - lo) % world == 0:
            # (the two collectives of one process group run in issue order: the gather reads what the scatter left)
            shard = self._shards.get(bi)
            if shard is None:
                shard = self._shards[bi] = torch.empty(((hi - lo) // world,), dtype=self.flat.dtype,
                                                       device=self.flat.device)
            rs = dist.reduce_scatter_tensor(shard, self.flat[lo:hi], group=self.group, async_op=True)
            self.issued.append("reduce_scatter_tensor")
            if dist.get_backend(self.group) == "nccl":
                # RCCL runs a communicator's collectives in issue order on its own stream: the gather can follow now
                # and both overlap the rest of backward
                self.works.append(rs)
                self.works.append(dist.all_gather_into_tensor(self.flat[lo:hi], shard, group=self.group,
                                                              async_op=True))
                self.issued.append("all_gather_into_tensor")
            else:  # gloo runs asynchronous work items concurrently: the gather is issued once the scatter is done
                self.deferred.append((rs, bi))
        else:  # "rccl", or a last bucket that does not divide into world shards
            self.works.append(dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))
            self.issued.append("all_reduce")

    def finish(self, average=True):
        for bi in range(len(self.buckets) - 1, -1, -1):
            if not self.launched[bi]:
                self._launch(bi)
        for rs, bi in self.deferred:
            rs.wait()
            lo, hi = self.buckets[bi]
            self.works.append(dist.all_gather_into_tensor(self.flat[lo:hi], self._shards[bi], group=self.group,
                                                          async_op=True))
            self.issued.append("all_gather_into_tensor")
        for w in self.works:
            w.wait()
        if average and self._active():
            self.flat.div_(dist.get_world_size(self.group))
        issued = self.issued
        self.reset()
        self.last_issued = issued


def synthetic_ground_truth(sample, seed=0, num_alpha_bins=12, map_size=(48, 48)):
    """Seeded synthetic labels for a `sample` of MonoPSRModel.build (benchmarks and tests; there is no dataset
    reader in this package): 3-D boxes near the proposal geometry, angle-bin labels through the reference's
    encoder, unit-scale local maps, a 70 % valid mask and a global map at the box position."""
    import numpy as np
    from monopsr_amd.core import orientation_encoder
    dev = sample['boxes_2d'].device
    B = sample['boxes_2d'].shape[0]
    rng = np.random.default_rng(seed)
    view = sample['est_view_angs'].reshape(-1).cpu().numpy().astype(np.float64)
    z = rng.uniform(8, 40, B)
    lwh = sample['mean_lwh'].cpu().numpy() + rng.normal(0, 0.1, (B, 3))
    boxes_3d = np.zeros((B, 7), np.float32)
    boxes_3d[:, 0], boxes_3d[:, 1], boxes_3d[:, 2] = z * np.tan(view), rng.uniform(1.2, 2.0, B), z
    boxes_3d[:, 3:6] = lwh
    alphas = rng.uniform(-np.pi, np.pi, B)
    boxes_3d[:, 6] = alphas + view
    enc = [orientation_encoder.np_orientation_to_angle_bin(a, num_alpha_bins, 0.0) for a in alphas]
    h, w = map_size
    xyz_local = rng.standard_normal((B, h, w, 3)).astype(np.float32)
    cen = np.stack([boxes_3d[:, 0], boxes_3d[:, 1] - boxes_3d[:, 5] / 2, boxes_3d[:, 2]], 1).astype(np.float32)
    t = lambda a, dt=torch.float32: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=dev)
    gt = dict(boxes_3d=t(boxes_3d), gt_alpha_bins=t([e[0] for e in enc], torch.int64),
              gt_alpha_regs=t(np.stack([e[1] for e in enc])), gt_alpha_valid_bins=t(np.stack([e[2] for e in enc])),
              gt_view_angs=t(view), gt_inst_xyz_maps_local=t(xyz_local),
              gt_inst_xyz_maps_global=t(xyz_local + cen[:, None, None, :]),
              gt_valid_mask_maps=t((rng.uniform(size=(B, h, w, 1)) > 0.3).astype(np.float32)))
    return gt


class _ConstantLr:
    optimizer_type = 'adam_optimizer'

    def __init__(self, lr):
        self.adam_optimizer = self
        self.learning_rate_type, self.learning_rate = 'constant_learning_rate', lr
        self.use_moving_average = False


class InstanceTrainer:
    """sample keys for step(): those of MonoPSRModel.build (rgb_image_crops, full_img_feature_crop, boxes_2d, cam_p,
    est_view_angs, class_indices, mean_lwh, prop_cen_z_offset) plus the ground truth the reference feeds through
    placeholders: boxes_3d (B,7), gt_alpha_bins (B), gt_alpha_regs (B,nb), gt_alpha_valid_bins (B,nb),
    gt_view_angs (B), gt_inst_xyz_maps_local / gt_inst_xyz_maps_global (B,h,w,3), gt_valid_mask_maps (B,h,w,1)."""

    def __init__(self, net, model_config, dataset_config, train_config=None, group=None, lr=None, clip_norm=1.0,
                 bucket_bytes=64 << 20, classes_name='Car', allreduce="rccl", force_collectives=False):
        """force_collectives: run the gradient exchange even on a one-rank process group (ReverseBucketReducer
        force_active): the real RCCL calls of an N > 1 step on a single-GPU box."""
        self.net, self.model_config, self.dataset_config = net, model_config, dataset_config
        self.clip_norm, self.classes_name = clip_norm, classes_name
        self.model = MonoPSRModel(model_config, dataset_config, net, 'train', classes_name, fused_heads=False)
        if lr is not None or train_config is None:
            self.optimizer = optimizer_builder.build(_ConstantLr(8e-5 if lr is None else lr))
        else:
            self.optimizer = optimizer_builder.build(train_config.optimizer)
        self.global_step = 0
        self._clip = None
        self.fused_update = True  # clip + Adam + moving average as one pass (False: the three separate passes, A/B / tests)
        spans = []
        base = net.grads.data_ptr()
        for L in net.layers:  # everything a layer's backward deposits: weight gradient and bias / beta gradient
            spans.append([((t.data_ptr() - base) // 4, (t.data_ptr() - base) // 4 + t.numel())
                          for t in (L.dw, L.db) if t is not None])
        self.spans = spans
        self.reducer = ReverseBucketReducer(net.grads, spans, bucket_bytes, group, mode=allreduce,
                                            force_active=force_collectives)
        for li, L in enumerate(net.layers):
            L.on_grad_ready = (lambda i=li: self.reducer.layer_ready(i))
        # a rank that exchanges over RCCL: the weight-gradient stream must not share a hardware queue with the communicator's
        # stream (every rank makes the same probe calls); the report travels in the bench line
        self.hardware_queues = None
        if self.reducer._active() and net.grads.is_cuda and dist.get_backend(group) == "nccl":
            from monopsr_amd.core import autograd_ops
            if autograd_ops.WGRAD_SIDE_STREAM and not os.environ.get("MPSR_DEBUG_NO_QUEUE_PROBE"):
                try:
                    self.hardware_queues = autograd_ops.settle_wgrad_stream(net.grads.device, group)
                except Exception as e:  # (a diagnostic and a tuning step: never the reason a trainer cannot be built)
                    self.hardware_queues = {"error": repr(e)[:300]}

    def forward(self, sample):
        out, _ = self.model.build(sample)
        return out

    def loss(self, out, sample):
        """-> (losses_dict, total_loss) of monopsr_model.py:554-958 for the outputs just built."""
        return self.model.loss(out, self.model.gt_dict, sample.get('gt_alpha_valid_bins'))

    def _clip_table(self, chunk=16384):
        """Chunk table of mpsr_clip_by_norm_segments: every weight / bias gradient is one variable (chunks of a variable
        consecutive, variables ascending).  -> (chunk_seg, chunk_begin, chunk_len, sumsq scratch of n_variables + n_chunks
        floats, n_variables)."""
        base = self.net.grads.data_ptr()
        seg, begin, length = [], [], []
        nseg = 0
        for L in self.net.layers:
            for t in (L.dw, L.db):
                if t is None:
                    continue
                lo, n = (t.data_ptr() - base) // 4, t.numel()
                for o in range(0, n, chunk):
                    seg.append(nseg)
                    begin.append(lo + o)
                    length.append(min(chunk, n - o))
                nseg += 1
        dev = self.net.grads.device
        return (torch.tensor(seg, dtype=torch.int32, device=dev), torch.tensor(begin, dtype=torch.int64, device=dev),
                torch.tensor(length, dtype=torch.int32, device=dev),
                torch.empty((nseg + len(seg),), dtype=torch.float32, device=dev), nseg)

    def clip_per_variable(self):
        """tf.clip_by_norm(g, clip_norm) per variable (weights and biases separately), on the reduced gradients:
        three launches over the flat buffer (deterministic norms)."""
        if self._clip is None:
            self._clip = self._clip_table()
        seg, begin, length, sumsq, nseg = self._clip
        _lib.check(_lib.lib().mpsr_clip_by_norm_segments(
            _lib.ptr(self.net.grads), _lib.ptr(seg), _lib.ptr(begin), _lib.ptr(length), seg.numel(), _lib.ptr(sumsq),
            sumsq.numel(), nseg, float(self.clip_norm), _lib.stream()))
        return self.net.grads

    def step(self, sample):
        """One training step, launched eagerly on the current stream.  (Rounds 4-5 also offered the step as ONE captured
        HIP graph; it was removed in r06: it no longer beat the eager step -- 52.2 vs 51.1 ms, the eager launches are
        back to back -- and replays that followed a device-wide synchronisation returned non-finite weight gradients
        in 10-30 % of the cases on this ROCm 7.2 / torch 2.10 stack for a reason that was never found, while eager steps
        with every uninitialised read made deterministic are clean: tests/test_poisoned_scratch_gpu.py.  DESIGN.md 4.6.)"""
        self.net.zero_grad()
        out = self.forward(sample)
        self.losses_dict, loss = self.loss(out, sample)
        bank = getattr(self.net, "dgrad_bank", None)
        if bank is not None:
            bank.refresh()  # the weights are this step's until apply_gradients: one launch packs every layer
        try:
            loss.backward()
        finally:
            if bank is not None:
                bank.invalidate()
        self.reducer.finish(average=True)
        if self.fused_update:
            # clip (after the reduce: the reference's order, core/trainer.py:76-81) -> Adam -> moving average in two
            # launches; net.grads keeps the reduced, UNclipped gradient (clip_per_variable() clips it in place)
            if self._clip is None:
                self._clip = self._clip_table()
            self.optimizer.apply_clipped_gradients(self.net, self.global_step, self._clip, self.clip_norm)
        else:
            if self.clip_norm:
                self.clip_per_variable()
            self.optimizer.apply_gradients(self.net, self.global_step)
        self.global_step += 1
        return loss.detach()

    # ------------------------------------------------------------------ checkpoint / resume (core/trainer.py:85,149-185)
    def save(self, checkpoint_dir, name='monopsr'):
        """Write the training state as a TensorFlow-format checkpoint `<dir>/<name>-<global_step, 8 digits>` (the
        naming of the reference's Saver(pad_step_number=True)) and update the directory's `checkpoint` state file.
        The variables are this trainer's FLAT buffers (BatchNorm already folded into the weights), so the file
        resumes this trainer; it is not interchangeable with a checkpoint of the reference's TF variables."""
        import os
        import numpy as np
        from monopsr_amd.core import tf_checkpoint
        net, opt = self.net, self.optimizer
        tensors = {
            'global_step': np.asarray(self.global_step, np.int64),
            'monopsr_amd/flat_params': net.params.cpu().numpy(),
            'monopsr_amd/flat_params/Adam': net.adam_m.cpu().numpy(),
            'monopsr_amd/flat_params/Adam_1': net.adam_v.cpu().numpy(),
            'monopsr_amd/adam_step': np.asarray(net.step_count, np.int64),
        }
        if opt.shadow is not None:
            tensors['monopsr_amd/flat_params/ExponentialMovingAverage'] = opt.shadow.cpu().numpy()
        for i, L in enumerate(net.layers):  # training-mode BatchNorm: the moving statistics are state, not parameters
            if L.batch_norm is not None:
                tensors['monopsr_amd/layer_%03d/BatchNorm/moving_mean' % i] = L.batch_norm.moving_mean.cpu().numpy()
                tensors['monopsr_amd/layer_%03d/BatchNorm/moving_variance' % i] = \
                    L.batch_norm.moving_variance.cpu().numpy()
        prefix = os.path.join(checkpoint_dir, '%s-%08d' % (name, self.global_step))
        tf_checkpoint.write_checkpoint(prefix, tensors)
        tf_checkpoint.write_checkpoint_state(checkpoint_dir, os.path.basename(prefix))
        return prefix

    def restore(self, path):
        """Resume from `save`'s output (a prefix, or a directory holding a `checkpoint` state file)."""
        from monopsr_amd.core import tf_checkpoint
        t = tf_checkpoint.read_checkpoint(path)
        net = self.net
        if t['monopsr_amd/flat_params'].shape != tuple(net.params.shape):
            raise ValueError('checkpoint holds %d parameters, this net %d' % (t['monopsr_amd/flat_params'].size,
                                                                          net.params.numel()))
        dev = net.params.device
        net.params.copy_(torch.from_numpy(t['monopsr_amd/flat_params']).to(dev))
        net.adam_m.copy_(torch.from_numpy(t['monopsr_amd/flat_params/Adam']).to(dev))
        net.adam_v.copy_(torch.from_numpy(t['monopsr_amd/flat_params/Adam_1']).to(dev))
        net.step_count = int(t['monopsr_amd/adam_step'])
        self.global_step = int(t['global_step'])
        for i, L in enumerate(net.layers):
            if L.batch_norm is not None:
                L.batch_norm.moving_mean.copy_(torch.from_numpy(
                    t['monopsr_amd/layer_%03d/BatchNorm/moving_mean' % i]).to(dev))
                L.batch_norm.moving_variance.copy_(torch.from_numpy(
                    t['monopsr_amd/layer_%03d/BatchNorm/moving_variance' % i]).to(dev))
        ema = t.get('monopsr_amd/flat_params/ExponentialMovingAverage')
        opt = self.optimizer
        if ema is not None and opt.shadow is not None and tuple(opt.shadow.shape) == tuple(ema.shape):
            opt.shadow.copy_(torch.from_numpy(ema).to(dev))  # in place: anyone holding the tensor keeps seeing the average
        else:
            opt.shadow = None if ema is None else torch.from_numpy(ema).to(dev)
        return self.global_step
