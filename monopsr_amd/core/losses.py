"""The two object-detection-API losses MonoPSR's loss builder instantiates (object_detection/core/losses.py:40-157,
283-317), with the same class names and call convention: loss(prediction_tensor, target_tensor, weights=...).
Inputs here are the heads' (1, num_boxes, k) tensors -- a few KB -- so these are plain torch expressions on the
device (autograd supplies the gradients); the map-sized terms go through the HIP kernels in losses_custom.
"""
import torch
import torch.nn.functional as F


class Loss(object):
    """object_detection/core/losses.py:40-89."""

    def __call__(self, prediction_tensor, target_tensor, ignore_nan_targets=False, scope=None, **params):
        if ignore_nan_targets:
            target_tensor = torch.where(torch.isnan(target_tensor), prediction_tensor, target_tensor)
        return self._compute_loss(prediction_tensor, target_tensor, **params)

    def _compute_loss(self, prediction_tensor, target_tensor, **params):
        raise NotImplementedError


def huber_elementwise(prediction_tensor, target_tensor, delta):
    """tf.losses.huber_loss term: 0.5 q^2 + delta (|e| - q) with q = min(|e|, delta)."""
    e = (prediction_tensor - target_tensor).abs()
    q = torch.clamp(e, max=delta)
    return 0.5 * q * q + delta * (e - q)


class WeightedSmoothL1LocalizationLoss(Loss):
    """losses.py:118-157: huber * weights with Reduction.NONE, summed over the code axis (2)."""

    def __init__(self, delta=1.0):
        self._delta = delta

    def _compute_loss(self, prediction_tensor, target_tensor, weights):
        if weights.dim() == prediction_tensor.dim() - 1:
            weights = weights.unsqueeze(-1)
        return (huber_elementwise(prediction_tensor, target_tensor, self._delta) * weights).sum(2)


class WeightedSoftmaxClassificationLoss(Loss):
    """losses.py:283-317: softmax cross entropy per row against (soft) targets, reshaped like weights, * weights."""

    def __init__(self, logit_scale=1.0):
        self._logit_scale = logit_scale

    def _compute_loss(self, prediction_tensor, target_tensor, weights):
        num_classes = prediction_tensor.shape[-1]
        logits = (prediction_tensor / self._logit_scale).reshape(-1, num_classes)
        ce = -(target_tensor.reshape(-1, num_classes) * F.log_softmax(logits, dim=1)).sum(1)
        return ce.reshape(weights.shape) * weights
