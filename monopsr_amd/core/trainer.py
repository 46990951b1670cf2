"""One data-parallel training step of the instance path: forward -> losses -> backward (gradients land in the flat
buffer) -> bucketed RCCL all-reduce overlapped with the rest of backward -> per-variable norm clip -> fused Adam.

What it mirrors of the reference's training setup (core/trainer.py:58-81, builders/optimizer_builder.py:61-112,
configs/monopsr_model_000.yaml:100-152): Adam at lr 8e-5, per-variable clip_by_norm(1.0) as
slim.learning.create_train_op(clip_gradient_norm=1.0) applies it, loss = weighted sum of per-output losses.
What it does not: exponential lr decay schedule bookkeeping, the EMA shadow variables, TF summaries, checkpoints.
Losses here: Chamfer on the local xyz map (losses_custom.ChamferDistance) and smooth-L1 on lwh / centroid outputs
(the reference's default regression loss type for those outputs).
"""
import torch
import torch.distributed as dist

from monopsr_amd.core import constants
from monopsr_amd.core import losses_custom
from monopsr_amd.core.models.monopsr import monopsr_output_builder


class ReverseBucketReducer:
    """All-reduce of a flat gradient buffer in buckets, launched from the END of the buffer towards the start as
    layers report their gradients ready -- backward visits layers in roughly reverse buffer order (heads, decoder,
    trunk), so the first buckets are on the wire while the trunk is still back-propagating."""

    def __init__(self, flat, layer_spans, bucket_bytes=64 << 20, group=None):
        self.flat, self.group = flat, group
        n = max(1, bucket_bytes // flat.element_size())
        self.buckets = [(lo, min(lo + n, flat.numel())) for lo in range(0, flat.numel(), n)]
        # layers overlapping each bucket
        self.members = [set() for _ in self.buckets]
        self.layer_buckets = []
        for li, (lo, hi) in enumerate(layer_spans):
            ids = [bi for bi, (blo, bhi) in enumerate(self.buckets) if lo < bhi and hi > blo]
            self.layer_buckets.append(ids)
            for bi in ids:
                self.members[bi].add(li)
        self.reset()

    def reset(self):
        self.pending = [set(m) for m in self.members]
        self.launched = [False] * len(self.buckets)
        self.works = []

    def _active(self):
        return dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1

    def layer_ready(self, li):
        for bi in self.layer_buckets[li]:
            self.pending[bi].discard(li)
            if not self.pending[bi] and not self.launched[bi]:
                self._launch(bi)

    def _launch(self, bi):
        self.launched[bi] = True
        if self._active():
            lo, hi = self.buckets[bi]
            self.works.append(dist.all_reduce(self.flat[lo:hi], group=self.group, async_op=True))

    def finish(self, average=True):
        for bi in range(len(self.buckets) - 1, -1, -1):
            if not self.launched[bi]:
                self._launch(bi)
        for w in self.works:
            w.wait()
        if average and self._active():
            self.flat.div_(dist.get_world_size(self.group))
        self.reset()


def smooth_l1(pred, target, delta=1.0):
    d = (pred - target).abs()
    return torch.where(d < delta, 0.5 * d * d, delta * (d - 0.5 * delta)).sum()


class InstanceTrainer:
    def __init__(self, net, model_config, dataset_config, group=None, lr=8e-5, clip_norm=1.0,
                 bucket_bytes=64 << 20, classes_name='Car'):
        self.net, self.model_config, self.dataset_config = net, model_config, dataset_config
        self.lr, self.clip_norm, self.classes_name = lr, clip_norm, classes_name
        spans = []
        for L in net.layers:
            lo = L.w.data_ptr() - net.params.data_ptr()
            spans.append((lo // 4, lo // 4 + L.w.numel()))
        self.spans = spans
        self.reducer = ReverseBucketReducer(net.grads, spans, bucket_bytes, group)
        for li, L in enumerate(net.layers):
            L.on_grad_ready = (lambda i=li: self.reducer.layer_ready(i))
        self.chamfer = losses_custom.ChamferDistance()

    def forward(self, sample):
        net = self.net
        feat = net.trunk(sample['rgb_image_crops'])
        feat_box, feat_map, xyz = net.squash_decoder(feat, sample['full_img_feature_crop'],
                                                     tuple(self.model_config.map_roi_size))
        features = {constants.FEATURES_FOR_MAP: feat_map, constants.FEATURES_FOR_BOX_3D: feat_box,
                    '_' + constants.KEY_INST_XYZ_MAP_LOCAL: xyz}
        B = feat_box.shape[0]
        b = monopsr_output_builder.MonoPSROutputBuilder(
            self.model_config.output_config, self.model_config, self.dataset_config, features, B,
            self.model_config.map_roi_size, sample['cam_p'].reshape(3, 4), train_val_test='test', device_net=net)
        out = b.get_output_dict()
        view = sample['est_view_angs'].reshape(-1, 1)
        shape = list(self.model_config.image_input_shape)
        b.add_inst_xyz_maps_local(None)
        b.add_proposal_fc_features(sample['boxes_2d'], view, sample['class_indices'], shape)
        f = b.get_proposal_fc_features()
        b.add_lwh_output(f, sample['mean_lwh'], None)
        b.add_alpha_output(f, None, None)
        b.add_view_ang_output(constants.KEY_VIEW_ANG, f, view, None)
        pz = b.get_prop_cen_z(sample['boxes_2d'], sample['prop_cen_z_offset'])
        py = b.get_prop_cen_y(sample['boxes_2d'], pz, self.classes_name)
        b.add_regression_fc_features(sample['boxes_2d'], view, sample['class_indices'], shape,
                                     out[constants.KEY_LWH + '_offs'], out[constants.KEY_ALPHA_BINS],
                                     out[constants.KEY_ALPHA_REGS], py, pz,
                                     self.dataset_config.obj_filter_config.depth_range[1])
        r = b.get_regression_fc_features()
        b.add_cen_y_output(constants.KEY_CEN_Y, r, py, None)
        b.add_cen_z_output(constants.KEY_CEN_Z, r, pz, None)
        b.add_cen_x_output(constants.KEY_CEN_X, out[constants.KEY_CEN_Z], out[constants.KEY_VIEW_ANG])
        b.add_centroids_output(constants.KEY_CENTROIDS, out[constants.KEY_CEN_X], out[constants.KEY_CEN_Y],
                               out[constants.KEY_CEN_Z], None)
        return b.get_output()

    def loss(self, out, gt):
        """gt: 'xyz' (B,h,w,3), 'mask' (B,h,w,1), 'lwh' (B,3), 'centroids' (B,3)."""
        B = out[constants.KEY_LWH].shape[0]
        total = self.chamfer(out[constants.KEY_INST_XYZ_MAP_LOCAL], gt['xyz'], gt['mask'])
        total = total + smooth_l1(out[constants.KEY_LWH], gt['lwh']) / B
        total = total + 0.1 * smooth_l1(out[constants.KEY_CENTROIDS], gt['centroids']) / B
        return total

    def clip_per_variable(self):
        """tf.clip_by_norm(g, clip_norm) per variable (weights and biases separately), on the reduced gradients."""
        g = self.net.grads
        for L in self.net.layers:
            for t in (L.dw, L.db):
                if t is None:
                    continue
                n = torch.linalg.vector_norm(t)
                t.mul_(torch.clamp(self.clip_norm / (n + 1e-30), max=1.0))
        return g

    def step(self, sample, gt):
        self.net.zero_grad()
        out = self.forward(sample)
        loss = self.loss(out, gt)
        loss.backward()
        self.reducer.finish(average=True)
        if self.clip_norm:
            self.clip_per_variable()
        self.net.adam_step(lr=self.lr)
        return loss.detach()
