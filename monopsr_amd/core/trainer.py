"""One data-parallel training step of the instance path: forward -> losses -> backward (gradients land in the flat
buffer) -> bucketed RCCL all-reduce overlapped with the rest of backward -> per-variable norm clip -> fused Adam.

What it mirrors of the reference's training setup (core/trainer.py:58-81, builders/optimizer_builder.py:24-122,
configs/monopsr_model_000.yaml:100-152): model.build() in 'train' mode -> model.loss() (the configured weighted sum
of per-output losses) -> Adam with the exponential-decay learning rate and the parameter moving average ->
per-variable clip_by_norm(1.0) as slim.learning.create_train_op(clip_gradient_norm=1.0) applies it.
Not mirrored: TF summaries and the checkpoint schedule.
"""
import torch
import torch.distributed as dist

from monopsr_amd import _lib
from monopsr_amd.builders import optimizer_builder
from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel


class ReverseBucketReducer:
    """All-reduce of a flat gradient buffer in buckets, launched from the END of the buffer towards the start as
    layers report their gradients ready -- backward visits layers in roughly reverse buffer order (heads, decoder,
    trunk), so the first buckets are on the wire while the trunk is still back-propagating."""

    def __init__(self, flat, layer_spans, bucket_bytes=64 << 20, group=None, mode="rccl"):
        """layer_spans[i]: the element range(s) of `flat` layer i's backward writes -- one (lo, hi) or a list of them
        (weight gradient AND bias gradient: a bucket holding any part of either must wait for the layer).
        mode: "rccl" = one all_reduce per bucket (the library's own algorithm choice); "direct" = reduce-scatter of the
        bucket into 1 / world shards followed by an all-gather (on the fully connected xGMI mesh every rank
        then sends each peer its shard ONCE per phase over that peer's own link, instead of 2 (world - 1) ring steps
        bound by one link).  Same sums up to the reduction order; selectable with bench.py --allreduce."""
        if mode not in ("rccl", "direct"):
            raise ValueError("reducer mode must be 'rccl' or 'direct'")
        self.flat, self.group, self.bucket_bytes, self.mode = flat, group, bucket_bytes, mode
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

    def _active(self):
        return (self.enabled and dist.is_available() and dist.is_initialized()
                and dist.get_world_size(self.group) > 1)

    def layer_ready(self, li):
        for bi in self.layer_buckets[li]:
            self.pending[bi].discard(li)
            if not self.pending[bi] and not self.launched[bi]:
                self._launch(bi)

    def _launch(self, bi):
        self.launched[bi] = True
        if self._active():
            lo, hi = self.buckets[bi]
            world = dist.get_world_size(self.group)
            if self.mode == "direct" and (hi - lo) % world == 0:
                # (the two collectives of one process group run in issue order: the gather reads what the scatter left)
                shard = self._shards.get(bi)
                if shard is None:
                    shard = self._shards[bi] = torch.empty(((hi - lo) // world,), dtype=self.flat.dtype,
                                                           device=self.flat.device)
                rs = dist.reduce_scatter_tensor(shard, self.flat[lo:hi], group=self.group, async_op=True)
                if dist.get_backend(self.group) == "nccl":
                    # RCCL runs a communicator's collectives in issue order on its own stream: the gather can follow now
                    # and both overlap the rest of backward
                    self.works.append(rs)
                    self.works.append(dist.all_gather_into_tensor(self.flat[lo:hi], shard, group=self.group,
                                                                  async_op=True))
                else:  # gloo runs asynchronous work items concurrently: the gather is issued once the scatter is done
                    self.deferred.append((rs, bi))
            else:  # "rccl", or a last bucket that does not divide into world shards
                self.works.append(dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))

    def finish(self, average=True):
        for bi in range(len(self.buckets) - 1, -1, -1):
            if not self.launched[bi]:
                self._launch(bi)
        for rs, bi in self.deferred:
            rs.wait()
            lo, hi = self.buckets[bi]
            self.works.append(dist.all_gather_into_tensor(self.flat[lo:hi], self._shards[bi], group=self.group,
                                                          async_op=True))
        for w in self.works:
            w.wait()
        if average and self._active():
            self.flat.div_(dist.get_world_size(self.group))
        self.reset()


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


class _StepGraph:
    """InstanceTrainer.capture_step(): eager warm-up calls, one captured call, then replays (see there)."""
    TICKS = 96  # eager launches on the caller's stream in front of every call (step())

    def __init__(self, trainer, warmup):
        self.tr, self.left = trainer, max(1, int(warmup))
        dev = trainer.net.params.device
        self.side = torch.cuda.Stream(device=dev)  # capture needs a non-default stream; the scratch caches are per stream
        self.lr_t = torch.zeros((), dtype=torch.float32, device=dev)
        self._tick = torch.zeros((), dtype=torch.float32, device=dev)  # (see step(): eager launches in front of every call)
        self.graph = self.static = self.loss = None

    def _copy_in(self, sample):
        for k, v in sample.items():
            if torch.is_tensor(v):
                dst = self.static[k]
                if dst.shape != v.shape or dst.dtype != v.dtype:
                    raise _lib.InvalidArgumentError("captured step: sample[%r] is %s %s, the captured call saw %s %s" %
                                                    (k, tuple(v.shape), v.dtype, tuple(dst.shape), dst.dtype))
                if dst.data_ptr() != v.data_ptr():
                    dst.copy_(v, non_blocking=True)
            elif self.static.get(k) is not v and self.static.get(k) != v:
                raise _lib.InvalidArgumentError("captured step: sample[%r] changed (only tensors may)" % k)

    def step(self, sample):
        tr = self.tr
        main = torch.cuda.current_stream()
        # Workaround, measured on this stack (ROCm 7.2, torch 2.10+rocm7.0; tools/graph_replay_probe.py): a replay of this
        # ~1100-node graph that directly follows a device-wide synchronisation (a .cpu(), a checkpoint save / restore, an
        # evaluation pass) after earlier replays came back with its FORWARD intact and scattered weight gradients
        # non-finite in 10-30 % of the cases -- whichever kernels the step uses (every A/B switch of the library tried),
        # eager steps never; a synchronisation or a spin kernel in front of the replay changes nothing.  Eager launches on
        # the caller's stream in front of EVERY call of this method -- the warm-up calls and the capturing call included:
        # in front of the replays alone they do not help (9 bad of 40) -- make it go away: 0 bad of 280 over the
        # scenarios that failed (8 launches: 3 of 40; 64: 0 of 40).  The cause sits below this package, in how the graph
        # is captured / launched around a synchronisation; the launches cost ~0.2 ms per step.
        for _ in range(self.TICKS):
            self._tick.add_(0.0)
        self.side.wait_stream(main)
        with torch.cuda.stream(self.side):
            if self.graph is None and self.left > 0:  # eager, on the capturing stream
                self.left -= 1
                loss = tr._eager_step(sample)
            elif self.graph is None:
                self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in sample.items()}
                self.lr_t.fill_(tr.optimizer.lr_t_of(tr.net, tr.global_step))
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph, stream=self.side):
                    self.loss = tr._eager_step(self.static, lr_t_dev=self.lr_t)
                # the graph's launches hold the addresses of this stream's scratch buffers: keep the tensors, so that a
                # cache eviction (more than 8 streams using the package) cannot hand the memory to someone else
                from monopsr_amd.core import device_net as dn
                self.pinned = dn.pinned_stream_scratch(tr.net.params.device, self.side)
                # (capture records, it does not run: this call's step is the first replay)
                self.graph.replay()
                loss = self.loss.clone()
            else:
                self._copy_in(sample)
                self.lr_t.fill_(tr.optimizer.lr_t_of(tr.net, tr.global_step))
                self.graph.replay()
                tr.global_step += 1
                loss = self.loss.clone()
        main.wait_stream(self.side)
        return loss


class InstanceTrainer:
    """sample keys for step(): those of MonoPSRModel.build (rgb_image_crops, full_img_feature_crop, boxes_2d, cam_p,
    est_view_angs, class_indices, mean_lwh, prop_cen_z_offset) plus the ground truth the reference feeds through
    placeholders: boxes_3d (B,7), gt_alpha_bins (B), gt_alpha_regs (B,nb), gt_alpha_valid_bins (B,nb),
    gt_view_angs (B), gt_inst_xyz_maps_local / gt_inst_xyz_maps_global (B,h,w,3), gt_valid_mask_maps (B,h,w,1)."""

    def __init__(self, net, model_config, dataset_config, train_config=None, group=None, lr=None, clip_norm=1.0,
                 bucket_bytes=64 << 20, classes_name='Car', allreduce="rccl"):
        self.net, self.model_config, self.dataset_config = net, model_config, dataset_config
        self.clip_norm, self.classes_name = clip_norm, classes_name
        self.model = MonoPSRModel(model_config, dataset_config, net, 'train', classes_name, fused_heads=False)
        if lr is not None or train_config is None:
            self.optimizer = optimizer_builder.build(_ConstantLr(8e-5 if lr is None else lr))
        else:
            self.optimizer = optimizer_builder.build(train_config.optimizer)
        self.global_step = 0
        self._clip = None
        self._graph = None
        spans = []
        base = net.grads.data_ptr()
        for L in net.layers:  # everything a layer's backward deposits: weight gradient and bias / beta gradient
            spans.append([((t.data_ptr() - base) // 4, (t.data_ptr() - base) // 4 + t.numel())
                          for t in (L.dw, L.db) if t is not None])
        self.spans = spans
        self.reducer = ReverseBucketReducer(net.grads, spans, bucket_bytes, group, mode=allreduce)
        for li, L in enumerate(net.layers):
            L.on_grad_ready = (lambda i=li: self.reducer.layer_ready(i))

    def forward(self, sample):
        out, _ = self.model.build(sample)
        return out

    def loss(self, out, sample):
        """-> (losses_dict, total_loss) of monopsr_model.py:554-958 for the outputs just built."""
        return self.model.loss(out, self.model.gt_dict, sample.get('gt_alpha_valid_bins'))

    def _clip_table(self, chunk=16384):
        """Chunk table of mpsr_clip_by_norm_segments: every weight / bias gradient is one variable."""
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
                torch.empty((nseg,), dtype=torch.float32, device=dev))

    def clip_per_variable(self):
        """tf.clip_by_norm(g, clip_norm) per variable (weights and biases separately), on the reduced gradients:
        two launches over the flat buffer."""
        if self._clip is None:
            self._clip = self._clip_table()
        seg, begin, length, sumsq = self._clip
        _lib.check(_lib.lib().mpsr_clip_by_norm_segments(
            _lib.ptr(self.net.grads), _lib.ptr(seg), _lib.ptr(begin), _lib.ptr(length), seg.numel(), _lib.ptr(sumsq),
            sumsq.numel(), float(self.clip_norm), _lib.stream()))
        return self.net.grads

    def step(self, sample):
        if self._graph is not None:
            return self._graph.step(sample)
        return self._eager_step(sample)

    def _eager_step(self, sample, lr_t_dev=None):
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
        if self.clip_norm:
            self.clip_per_variable()
        if lr_t_dev is None:
            self.optimizer.apply_gradients(self.net, self.global_step)
        else:
            self.optimizer.apply_gradients_lr_dev(self.net, lr_t_dev)
        self.global_step += 1
        return loss.detach()

    def capture_step(self, warmup=2):
        """From now on step() replays ONE HIP graph: the ~1100 launches of a training step (forward, configured losses,
        backward, clip, Adam, moving average) captured once -- the eager step leaves 2-3 ms of launch gaps per step in
        its loss / head / optimizer sections, where the host issues hundreds of tiny kernels (`tools/train_bench.py
        --graph`: 59.8 -> 56.8 ms).  The first `warmup` + 1 calls run eagerly (they size the per-stream scratch caches,
        create the moving average and are the captured call); every later call copies the sample into the captured
        call's tensors, writes the step's learning rate where the captured Adam launch reads it and replays.  Shapes
        and dtypes of the sample must stay those of the first call.  Single-process training only: a data-parallel
        step exchanges gradients through torch.distributed while backward runs and stays eager.
        Two things to know before choosing it (r05): it no longer beats the eager step in time (52.2 vs 51.1 ms: the
        eager launches are back to back by now), and replays of a graph this large need the workaround in
        _StepGraph.step() on this ROCm / torch stack -- without it a replay that follows a device-wide synchronisation
        returned non-finite gradients in 10-30 % of the cases (tools/graph_replay_probe.py)."""
        if self.reducer._active():
            raise RuntimeError("capture_step(): data-parallel steps stay eager (the bucketed all-reduce is issued "
                               "from Python while backward runs)")
        if self._graph is None:
            self._graph = _StepGraph(self, warmup)
        return self

    def release_step_graph(self):
        self._graph = None

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
            # in place: a captured step graph (capture_step) has this tensor's address baked into its lerp launch --
            # rebinding the attribute would leave the replays averaging into a buffer nobody reads any more
            opt.shadow.copy_(torch.from_numpy(ema).to(dev))
        else:
            opt.shadow = None if ema is None else torch.from_numpy(ema).to(dev)
            # the moving average appeared, vanished or changed size: a captured graph no longer describes the step
            self.release_step_graph()
        return self.global_step
