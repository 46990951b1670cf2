"""The object-detection-API losses MonoPSR's loss builder instantiates (object_detection/core/losses.py:40-157,
223-317), with the same class names and call convention: loss(prediction_tensor, target_tensor, weights=...).
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
    """tf.losses.huber_loss term: 0.5 q^2 + delta (|e| - q) with q = min(|e|, delta), i.e. 0.5 e^2 for |e| <= delta and
    delta (|e| - 0.5 delta) beyond -- torch's huber_loss, ONE launch forward and one backward where the expression
    written out was 8 and ~12 on these few-KB tensors (r06 launch census of the training step)."""
    if prediction_tensor.shape != target_tensor.shape:
        prediction_tensor, target_tensor = torch.broadcast_tensors(prediction_tensor, target_tensor)
    return F.huber_loss(prediction_tensor, target_tensor, reduction='none', delta=float(delta))


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


class SigmoidFocalClassificationLoss(Loss):
    """losses.py:223-280 (loss type 'focal'): sigmoid cross entropy per entry, down-weighted by (1 - p_t)^gamma and
    balanced by alpha, times weights (batch, anchors) broadcast over the class axis.  As in the reference a gamma of
    0 / None and an alpha of None switch the respective factor off; `class_indices` restricts the loss to some
    classes."""

    def __init__(self, gamma=2.0, alpha=0.25):
        self._alpha = alpha
        self._gamma = gamma

    def _compute_loss(self, prediction_tensor, target_tensor, weights, class_indices=None):
        weights = weights.unsqueeze(2)
        if class_indices is not None:
            dense = torch.zeros(prediction_tensor.shape[2], dtype=prediction_tensor.dtype,
                                device=prediction_tensor.device)
            dense[torch.as_tensor(class_indices, dtype=torch.int64, device=prediction_tensor.device)] = 1.0
            weights = weights * dense.reshape(1, 1, -1)
        per_entry_cross_ent = F.binary_cross_entropy_with_logits(prediction_tensor, target_tensor, reduction='none')
        prediction_probabilities = torch.sigmoid(prediction_tensor)
        p_t = target_tensor * prediction_probabilities + (1 - target_tensor) * (1 - prediction_probabilities)
        loss = per_entry_cross_ent
        if self._gamma:
            loss = torch.pow(1.0 - p_t, self._gamma) * loss
        if self._alpha is not None:
            loss = (target_tensor * self._alpha + (1 - target_tensor) * (1 - self._alpha)) * loss
        return loss * weights
