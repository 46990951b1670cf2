"""Trainable form of the instance path: the same layers as DeviceNet, sequenced in Python over
the differentiable kernel wrappers of autograd_ops, with every parameter in ONE flat fp32 buffer and every
gradient in ONE flat buffer of the same layout -- the buffer the data-parallel all-reduce
(core/data_parallel.BucketedAllReduce) and the fused Adam kernel (mpsr_adam_step) run over.

Parameterisation: inference-mode BatchNorm is folded into the convolutions (core/weights.fold_conv), so the
trainable tensors are the folded weight and bias of each layer.  The trunks' BatchNorm is frozen in the reference
too (faster_rcnn_resnet_v1_feature_extractor.py:63,238); training the folded weight instead of (w, gamma, beta)
separately is a re-parameterisation of the same function, stated here because it is not the reference's variable set.
"""
import os

import numpy as np
import torch

from monopsr_amd import _lib
from monopsr_amd.core import autograd_ops as ops
from monopsr_amd.core import weights as W


def _im2col_root(img, kpad):
    img = img.contiguous()
    B, H, Wd, _ = img.shape
    oh, ow = (H + 6 - 7) // 2 + 1, (Wd + 6 - 7) // 2 + 1
    cols = torch.empty((B * oh * ow, 1, 1, kpad), dtype=torch.float32, device=img.device)
    _lib.check(_lib.lib().mpsr_im2col_root(_lib.ptr(img), B, H, Wd, _lib.ptr(cols), kpad, _lib.stream()))
    return cols, oh, ow


class TrainNet:
    def __init__(self, weights, device="cuda", width_div=1, with_heads=True, full_trunk=False, decoder_bn='frozen',
                 bn_group=None, dgrad_bank=True):
        """full_trunk: also hold (and train) the full-image ResNet-101 (`weights` must then carry both scopes); its
        layers follow the heads in the flat buffer, so the flat gradient is the reference's whole 100 M-parameter
        set (401 MB at full width)."""
        """decoder_bn: 'frozen' = the map decoder's BatchNorm runs on its moving statistics, folded into the
        convolutions like the trunks' (an inference-mode re-parameterisation); 'batch' = the reference's training
        graph (net_builder.py:76-87, is_training=True): batch statistics over the step's instances, beta trained,
        moving statistics updated with decay 0.999, statistics per rank in a data-parallel run; 'batch_global' = the
        same with the sums pooled over the ranks of `bn_group` (default group when None): the statistics of the whole
        step's batch, as the reference's single-process step has them -- a sharded batch then gives the unsharded
        batch's activations and gradients (2 C fp64 numbers per layer and direction on the wire)."""
        if decoder_bn not in ('frozen', 'batch', 'batch_global'):
            raise ValueError("decoder_bn must be 'frozen', 'batch' or 'batch_global'")
        self.decoder_bn = decoder_bn
        self.device = torch.device(device)
        if full_trunk and self.two_stream_trunks:
            # builders/net_builder.py runs the full-image branch on this stream next to the crop trunk (created on first
            # use).  Autograd runs a node's backward on the stream of its forward, so the two trunks' backward passes
            # overlap as well; their deposits are joined at the end of the pass (autograd_ops.join_wgrad_stream).
            self.side_stream = None
            # Up to 64 boxes.  Above, a third busy stream is a gamble on the runtime's four hardware queues: alone
            # (tools/train_bench.py --full-image) the own stream wins at every size -- 256 boxes 61.05 -> 60.1 ms, 128: 40.7 ->
            # 39.4, 32: 26.2 -> 24.5 -- but inside bench.py, where a dozen other streams and an RCCL communicator exist,
            # 256 boxes read 63.7 with it and 61.4 without (same box, twice).  MPSR_DEBUG_TRUNK_STREAM_MAX_BOXES overrides.
            self.side_stream_max_boxes = int(os.environ.get("MPSR_DEBUG_TRUNK_STREAM_MAX_BOXES", 64))
        parts = [W.pack_trunk(weights, W.CROP_SCOPE, width_div), W.pack_decoder(weights, width_div)]
        fc_names = [n for n, _, _, _ in W.head_fc_specs()] if with_heads else []
        blobs, recs, base = [], [], 0
        for blob, records in parts:
            for r in records:
                recs.append(dict(r, w_off=r["w_off"] + base, b_off=(r["b_off"] + base) if r["b_off"] >= 0 else -1))
            blobs.append(blob)
            base += blob.size
        self.fc_index = {}
        for name in fc_names:
            w = weights[name + "/weights"]  # (in, out)
            kpad = (w.shape[0] + 3) // 4 * 4
            w_ok = np.zeros((w.shape[1], kpad), np.float32)
            w_ok[:, :w.shape[0]] = w.T
            b = weights[name + "/biases"].astype(np.float32)
            self.fc_index[name] = (len(recs), w.shape[0])
            wsz = (w_ok.size + 63) // 64 * 64  # keep every tensor 256-byte aligned (16-byte loads in the kernels)
            bsz = (b.size + 63) // 64 * 64
            recs.append(dict(cin=kpad, cout=w.shape[1], kh=1, kw=1, dilation=1, relu=0, w_off=base,
                             b_off=base + wsz))
            blobs += [w_ok.ravel(), np.zeros(wsz - w_ok.size, np.float32), b, np.zeros(bsz - b.size, np.float32)]
            base += wsz + bsz
        self.full_base = None
        if full_trunk:
            blob, records = W.pack_trunk(weights, W.FULL_SCOPE, width_div)
            self.full_base = len(recs)
            for r in records:
                recs.append(dict(r, w_off=r["w_off"] + base, b_off=(r["b_off"] + base) if r["b_off"] >= 0 else -1))
            blobs.append(blob)
            base += blob.size
        flat = np.concatenate(blobs)
        self.params = torch.from_numpy(flat).to(self.device)
        self.grads = torch.zeros_like(self.params)
        self.adam_m = torch.zeros_like(self.params)
        self.adam_v = torch.zeros_like(self.params)
        self.step_count = 0
        self.layers = []
        for r in recs:
            n = r["cout"] * r["kh"] * r["kw"] * r["cin"]
            wv = self.params[r["w_off"]:r["w_off"] + n].view(r["cout"], -1)
            dwv = self.grads[r["w_off"]:r["w_off"] + n].view(r["cout"], -1)
            bv = self.params[r["b_off"]:r["b_off"] + r["cout"]] if r["b_off"] >= 0 else None
            dbv = self.grads[r["b_off"]:r["b_off"] + r["cout"]] if r["b_off"] >= 0 else None
            self.layers.append(ops.LayerRef(wv, bv, dwv, dbv, r["cin"], r["cout"], r["kh"], r["kw"], r["dilation"],
                                            bool(r["relu"])))
        self.n_trunk = len(parts[0][1])
        self.n_dec = len(parts[1][1])
        # data-gradient layouts of every layer, made by one launch per step (the trainer refreshes it before backward)
        # (dgrad_bank=False: each layer's layout is packed inside backward, one launch per layer -- the A/B of
        # tools/train_bench.py --dgrad-bank 0)
        # (the two root convolutions read images: no data gradient, no slice -- ADVICE r05)
        roots = [0] + ([self.full_base] if self.full_base is not None else [])
        self.dgrad_bank = (ops.DgradBank(self.layers, self.device, skip=roots)
                           if (dgrad_bank and self.device.type == "cuda") else None)
        if decoder_bn in ('batch', 'batch_global'):
            # decoder records: squash (2 GEMMs), then one record per spec; BatchNorm layers get the UNFOLDED kernel
            # in their weight slot and beta in their bias slot
            idx = self.n_trunk + 1
            for name, kh, kw, cin, cout, has_bn, has_bias, relu in W.scaled_decoder_specs(width_div):
                if name.startswith("squash"):
                    continue
                idx += 1
                if not has_bn:
                    continue
                L = self.layers[idx]
                w_ok, _ = W.fold_conv(weights[name + "/weights"])
                L.w.copy_(torch.from_numpy(w_ok).to(self.device))
                L.b.copy_(torch.from_numpy(weights[name + "/BatchNorm/beta"].astype(np.float32)).to(self.device))
                t = lambda a: torch.from_numpy(a.astype(np.float32)).to(self.device)
                L.batch_norm = ops.BatchNormState(t(weights[name + "/BatchNorm/moving_mean"]),
                                                  t(weights[name + "/BatchNorm/moving_variance"]), W.DECODER_BN_EPS,
                                                  sync_group=(bn_group if bn_group is not None else True)
                                                  if decoder_bn == 'batch_global' else None)

    # ------------------------------------------------------------------ forward pieces
    # full_trunk: the full-image trunk (6080 pixel rows: <= 190 tiles for 256 CUs) on its own stream at up to 64 boxes
    two_stream_trunks = True
    fused_units = True  # trunk(): bottleneck units as single autograd nodes (ops.BottleneckFn); False = layer by layer
    linked_units = True  # ... and chained: a unit's input gradient leaves through the previous unit's ReLU mask
    fused_upsampled_convs = True  # squash_decoder(): resize -> conv pairs as one tap-GEMM operator, forward and backward

    def trunk(self, img, scope='crop'):
        if scope == 'crop':
            L = self.layers
        elif scope == 'full' and self.full_base is not None:
            L = self.layers[self.full_base:]
        else:
            raise _lib.InvalidArgumentError("TrainNet was built without the %r trunk (full_trunk=True adds the "
                                            "full-image one)" % scope)
        B = img.shape[0]
        cols, oh, ow = _im2col_root(img, L[0].cin)
        x = ops.conv2d(cols, L[0]).reshape(B, oh, ow, L[0].cout)
        x = ops.max_pool(x, 3, 2, "SAME")
        li = 1
        link = None  # (the first unit reads the max-pool's output: no ReLU mask belongs to its input gradient)
        for blk, units in enumerate((3, 4, 23)):
            for u in range(units):
                shortcut = None
                if u == 0:
                    shortcut = L[li]
                    li += 1
                if self.fused_units:  # one autograd node per unit: the two gradients of its input meet in a conv epilogue
                    # consecutive units are linked: the tensor between them has no other reader, so a unit may hand
                    # its input gradient over already masked by the previous unit's ReLU (ops.UnitLink)
                    out_link = ops.UnitLink() if self.linked_units else None
                    x = ops.bottleneck(x, L[li], L[li + 1], L[li + 2], shortcut, link, out_link)
                    link = out_link
                else:
                    residual = ops.conv2d(x, shortcut) if shortcut is not None else x
                    t = ops.conv2d(x, L[li])
                    t = ops.conv2d(t, L[li + 1])
                    x = ops.conv2d(t, L[li + 2], residual=residual)
                li += 3
        return x

    def squash_decoder(self, crop_feat, full_feat, map_size=(48, 48)):
        L = self.layers[self.n_trunk:self.n_trunk + self.n_dec]
        part = ops.conv2d(crop_feat, L[0])
        sq = ops.conv2d(full_feat, L[1], residual=part)
        feat_box = ops.max_pool(sq, 2, 2, "VALID")
        half = (map_size[0] // 2, map_size[1] // 2)

        def up_conv(x, layer, size):  # resize -> conv: one operator where the tap-GEMM form applies (csrc/upconv.hip)
            if self.fused_upsampled_convs and ops.upsampled_conv_applies(tuple(x.shape), layer, size):
                return ops.upsampled_conv2d(x, layer, size, True)
            return ops.conv2d(ops.resize_bilinear(x, size, True), layer)
        y = ops.conv2d(up_conv(sq, L[2], half), L[3])
        feat_map = ops.conv2d(up_conv(y, L[4], tuple(map_size)), L[5])
        xyz = ops.conv2d(feat_map, L[6])
        return feat_box, feat_map, xyz

    # differentiable image operators the feature builder uses between the trunks (net_builder.py:54-60)
    def crop_and_resize(self, image, boxes, box_ind, crop_size, extrapolation_value=0.0):
        return ops.crop_and_resize(image, boxes, box_ind, crop_size, extrapolation_value)

    def max_pool(self, x, k, s, padding="VALID"):
        return ops.max_pool(x, k, s, padding)

    def fully_connected(self, x, name, relu):
        """Differentiable slim.fully_connected `name` (used by MonoPSROutputBuilder in training mode).  The flat
        parameter buffer holds the FC layers of monopsr_model_000's output set (core/weights.head_fc_specs); the
        other output types of the builder (alpha 'dc_rotation' with its own variable, view_ang 'offset', cen_z
        'direct', the predicted valid mask) are inference-only: asking for them here is an error, not a silent
        layer without gradients."""
        if name not in self.fc_index:
            raise _lib.InvalidArgumentError(
                "TrainNet has no trainable layer %r: it trains the output set of monopsr_model_000.yaml (%s); the "
                "output builder's other variants are inference-only (DeviceNet)" % (name, ", ".join(self.fc_index)))
        idx, fin = self.fc_index[name]
        L = self.layers[idx]
        L.relu = bool(relu)
        if x.shape[1] != fin:
            raise _lib.InvalidArgumentError("%s expects %d input features, got %d" % (name, fin, x.shape[1]))
        if L.cin != fin:
            x = torch.nn.functional.pad(x, (0, L.cin - fin))
        B = x.shape[0]
        return ops.conv2d(x.reshape(B, 1, 1, L.cin), L).reshape(B, -1)

    # ------------------------------------------------------------------ export
    def export_weights(self, width_div=1):
        """The trained parameters as a dict keyed by the reference's TF variable names (the layout
        core/weights.py documents, what DeviceNet / checkpoint_utils.save_checkpoint take).  Layers trained in folded
        form come back as an equivalent variable set: kernel = folded kernel, BatchNorm beta = folded bias,
        gamma = 1, moving_mean = 0, moving_variance = 1 - eps (so that inference-mode BatchNorm is the identity plus
        beta).  Decoder layers trained with batch statistics export their own kernel, beta and moving statistics."""
        out = {}

        def hwio(w_ok, kh, kw, cin):
            cout = w_ok.shape[0]
            return np.ascontiguousarray(w_ok.reshape(cout, kh, kw, cin).transpose(1, 2, 3, 0))

        def identity_bn(name, b, eps, with_gamma):
            c = b.shape[0]
            if with_gamma:
                out[name + "/BatchNorm/gamma"] = np.ones(c, np.float32)
            out[name + "/BatchNorm/beta"] = b
            out[name + "/BatchNorm/moving_mean"] = np.zeros(c, np.float32)
            out[name + "/BatchNorm/moving_variance"] = np.full(c, 1.0 - eps, np.float32)

        def trunk(scope, base):
            for i, sp in enumerate(W.scaled_trunk_specs(scope, width_div)):
                L = self.layers[base + i]
                w, b = L.w.detach().cpu().numpy(), L.b.detach().cpu().numpy()
                if sp["role"] == "root":
                    out[sp["name"] + "/weights"] = hwio(w[:, :W.ROOT_K], 7, 7, 3)
                else:
                    out[sp["name"] + "/weights"] = hwio(w, sp["kh"], sp["kw"], sp["cin"])
                identity_bn(sp["name"], b, W.TRUNK_BN_EPS, True)
        trunk(W.CROP_SCOPE, 0)
        if self.full_base is not None:
            trunk(W.FULL_SCOPE, self.full_base)
        idx = self.n_trunk
        for name, kh, kw, cin, cout, has_bn, has_bias, relu in W.scaled_decoder_specs(width_div):
            if name.startswith("squash"):
                a, b2 = self.layers[idx], self.layers[idx + 1]
                w = np.concatenate([a.w.detach().cpu().numpy(), b2.w.detach().cpu().numpy()], 1)
                out[name + "/weights"] = hwio(w, 1, 1, cin)
                out[name + "/biases"] = b2.b.detach().cpu().numpy()
                idx += 2
                continue
            L = self.layers[idx]
            idx += 1
            out[name + "/weights"] = hwio(L.w.detach().cpu().numpy(), kh, kw, cin)
            if not has_bn:
                out[name + "/biases"] = L.b.detach().cpu().numpy()
            elif L.batch_norm is not None:
                out[name + "/BatchNorm/beta"] = L.b.detach().cpu().numpy()
                out[name + "/BatchNorm/moving_mean"] = L.batch_norm.moving_mean.cpu().numpy()
                out[name + "/BatchNorm/moving_variance"] = L.batch_norm.moving_variance.cpu().numpy()
            else:
                identity_bn(name, L.b.detach().cpu().numpy(), W.DECODER_BN_EPS, False)
        for name, (i, fin) in self.fc_index.items():
            L = self.layers[i]
            out[name + "/weights"] = np.ascontiguousarray(L.w.detach().cpu().numpy()[:, :fin].T)
            out[name + "/biases"] = L.b.detach().cpu().numpy()
        return out

    # ------------------------------------------------------------------ optimisation
    def zero_grad(self):
        # (whatever an earlier backward pass left on the weight-gradient stream is ordered in front of the zeroing, and a
        # pass that ended in an exception cannot leave its end-of-pass join marked as queued)
        if self.grads.is_cuda:
            ops.join_wgrad_stream(self.device)
        self.grads.zero_()

    @staticmethod
    def adam_rate(lr, step, beta1=0.9, beta2=0.999):
        """lr_t of step `step` (1-based): what mpsr_adam_step computes from (lr, step) on the host."""
        import math
        return float(lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step))

    def adam_step_lr_dev(self, lr_t_dev, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """adam_step with the bias-corrected rate read from the device scalar `lr_t_dev` at run time (for a caller whose
        schedule lives on the device; `adam_rate` gives the value); the caller keeps step_count."""
        _lib.check(_lib.lib().mpsr_adam_step_lr_dev(_lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m),
                                                    _lib.ptr(self.adam_v), self.params.numel(), _lib.ptr(lr_t_dev),
                                                    beta1, beta2, eps, grad_scale, _lib.stream()))

    def clip_adam_ema_step(self, clip_table, clip_norm, lr=8e-5, shadow=None, ema_decay=0.0, beta1=0.9, beta2=0.999,
                           eps=1e-8):
        """Per-variable clip_by_norm + Adam + the parameter moving average as three launches over the chunk table
        `clip_table` = (chunk_seg, chunk_begin, chunk_len, sumsq, n_variables) of InstanceTrainer._clip_table (mpsr_clip_adam_ema_step:
        the gradient is scaled on its way into the update and NOT written back).  clip_norm 0 / None: no clipping."""
        seg, begin, length, sumsq, nseg = clip_table
        self.step_count += 1
        _lib.check(_lib.lib().mpsr_clip_adam_ema_step(
            _lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m), _lib.ptr(self.adam_v), _lib.ptr(shadow),
            _lib.ptr(seg), _lib.ptr(begin), _lib.ptr(length), seg.numel(), _lib.ptr(sumsq), sumsq.numel(),
            nseg, float(clip_norm or 0.0), lr, beta1, beta2, eps, self.step_count, float(ema_decay), _lib.stream()))

    def adam_step(self, lr=8e-5, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """tf.train.AdamOptimizer update (optimizer_builder.py:61-80; lr 8e-5 from monopsr_model_000.yaml:141-147)."""
        self.step_count += 1
        _lib.check(_lib.lib().mpsr_adam_step(_lib.ptr(self.params), _lib.ptr(self.grads), _lib.ptr(self.adam_m),
                                             _lib.ptr(self.adam_v), self.params.numel(), lr, beta1, beta2, eps,
                                             self.step_count, grad_scale, _lib.stream()))
