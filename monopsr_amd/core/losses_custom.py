"""The reference's custom losses (core/losses_custom.py), same class names and call convention, over the HIP ops:
point-cloud losses (:135-198), the masked smooth-L1 the maps are trained with (:93-132), berHu (:45-90) and the
sigmoid mask loss (:201-232)."""
import torch
import torch.nn.functional as F

from monopsr_amd import _lib
from monopsr_amd.core.losses import Loss
from monopsr_amd.tf_ops.approxmatch import tf_approxmatch
from monopsr_amd.tf_ops.nn_distance import tf_nndistance


class _HuberNonZero(torch.autograd.Function):
    """sum(huber(pred - target) * w) / #(broadcast w != 0) via mpsr_huber_loss_sums / mpsr_huber_loss_grad."""

    @staticmethod
    def forward(ctx, pred, target, weights, delta):
        b, c = pred.shape[0], pred.shape[-1]
        p = pred[0].numel() // c if b else 0
        pred, target = pred.contiguous().float(), target.contiguous().float()
        weights = weights.contiguous().float()
        both = torch.empty((2, b), dtype=torch.float32, device=pred.device)  # per-instance sums | non-zero counts
        _lib.check(_lib.lib().mpsr_huber_loss_sums(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(weights), b, p, c,
                                                   float(delta), both[0].data_ptr(), both[1].data_ptr(), _lib.stream()))
        tot = both.sum(1)
        n = torch.clamp(tot[1], min=1.0)  # (no non-zero weight: the sum is 0 too, 0 / 1 = the reference's 0)
        ctx.save_for_backward(pred, target, weights, n)
        ctx.meta = (b, p, c, float(delta))
        return tot[0] / n

    @staticmethod
    def backward(ctx, g):
        pred, target, weights, n = ctx.saved_tensors
        b, p, c, delta = ctx.meta
        scale = (g.float() / n).reshape(1)  # (all weights zero: the kernel multiplies by them, the gradient is 0)
        grad = torch.empty_like(pred)
        _lib.check(_lib.lib().mpsr_huber_loss_grad(_lib.ptr(pred), _lib.ptr(target), _lib.ptr(weights),
                                                   _lib.ptr(scale), b, p, c, delta, _lib.ptr(grad), _lib.stream()))
        return grad, None, None, None


class WeightedNonZeroSmoothL1LocalizationLoss(Loss):
    """losses_custom.py:93-132: tf.losses.huber_loss with Reduction.SUM_BY_NONZERO_WEIGHTS -> a scalar.
    weights must be prediction_tensor's shape with a last axis of 1 (a per-pixel / per-box mask)."""

    def __init__(self, delta=1.0):
        self._delta = delta

    def _compute_loss(self, prediction_tensor, target_tensor, weights):
        if tuple(weights.shape) != tuple(prediction_tensor.shape[:-1]) + (1,):
            raise _lib.InvalidArgumentError("weights must have shape prediction.shape[:-1] + (1,), got %s for %s"
                                            % (tuple(weights.shape), tuple(prediction_tensor.shape)))
        if tuple(target_tensor.shape) != tuple(prediction_tensor.shape):
            raise _lib.InvalidArgumentError("prediction and target shapes differ")
        return _HuberNonZero.apply(prediction_tensor, target_tensor, weights, self._delta)


class WeightedBerHu(Loss):
    """losses_custom.py:45-90 (not selected by model 000's loss_config; small torch expression)."""

    def _compute_loss(self, prediction_tensor, target_tensor, weights):
        error = prediction_tensor - target_tensor
        abs_error = error.abs()
        l1_thresh = abs_error.max() / 5.0
        per_pixel = torch.where(abs_error <= l1_thresh, abs_error, (error ** 2 + l1_thresh ** 2) / (2 * l1_thresh))
        weights = weights.float()
        num_valid = (weights != 0).sum().float()
        total = (per_pixel * weights).sum()
        return torch.where(num_valid > 0, total / torch.clamp(num_valid, min=1.0), torch.zeros_like(total))


class SigmoidClassificationLoss(Loss):
    """losses_custom.py:201-232: per-entry sigmoid cross entropy; the weights argument is ignored there too."""

    def _compute_loss(self, prediction_tensor, target_tensor, class_indices=None, weights=None):
        return F.binary_cross_entropy_with_logits(prediction_tensor, target_tensor, reduction='none')


def _masked_points(prediction_tensor, target_tensor, weights):
    # losses_custom.py:151-158 / 184-191: multiply by the valid mask, reshape (B,h,w,3) -> (B,h*w,3)
    batch_size = prediction_tensor.shape[0]
    pred = (prediction_tensor * weights).reshape(batch_size, -1, 3)
    tgt = (target_tensor * weights).reshape(batch_size, -1, 3)
    return batch_size, pred, tgt


class EarthMoversDistance:
    """Approximation of the Earth Mover's Distance that compares two point clouds."""

    def __call__(self, prediction_tensor, target_tensor, weights):
        batch_size, pred, tgt = _masked_points(prediction_tensor, target_tensor, weights)
        # match_cost(pred, tgt, approx_match(pred, tgt)) of the reference, fused: the (B, m, n) match tensor
        # (4.3 GB at 256 x 2048^2) is never stored, value and gradients are the same
        distances = tf_approxmatch.emd_cost(pred, tgt)
        return distances.sum() / float(batch_size)


class ChamferDistance:
    """Computes the chamfer distance between two point clouds."""

    def __call__(self, prediction_tensor, target_tensor, weights):
        batch_size, pred, tgt = _masked_points(prediction_tensor, target_tensor, weights)
        dist1, _, dist2, _ = tf_nndistance.nn_distance(pred, tgt)
        return (dist1.sum() + dist2.sum()) / float(batch_size)
