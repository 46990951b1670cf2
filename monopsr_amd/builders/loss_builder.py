"""Loss selection by config: same entry points as the reference's builders/loss_builder.py:8-85
(get_loss_type_and_weight, build_loss, add_loss_tensor), driven by a registry instead of an if-chain."""
import torch

from monopsr_amd.core import losses, losses_custom

# loss type string (model_config.loss_config entries, yaml:102-117) -> constructor
_REGISTRY = {
    'smooth_l1': losses.WeightedSmoothL1LocalizationLoss,
    'smooth_l1_nonzero': losses_custom.WeightedNonZeroSmoothL1LocalizationLoss,
    'softmax': losses.WeightedSoftmaxClassificationLoss,
    'softmax_temp': lambda: losses.WeightedSoftmaxClassificationLoss(0.5),
    'focal': losses.SigmoidFocalClassificationLoss,
    'sigmoid_ce': losses_custom.SigmoidClassificationLoss,
    'berHu': losses_custom.WeightedBerHu,
    'chamfer_dist': losses_custom.ChamferDistance,
    'emd': losses_custom.EarthMoversDistance,
}


def get_loss_type_and_weight(loss_config, output_rep):
    """-> (type string, weight) of `output_rep`'s entry, e.g. ['smooth_l1', 0.1]."""
    entry = getattr(loss_config, output_rep, None)
    if entry is None:
        raise ValueError('Loss not configured for output_rep:', output_rep)
    return entry[0], entry[1]


def build_loss(loss_type):
    if loss_type not in _REGISTRY:
        raise ValueError('Invalid loss type', loss_type)
    return _REGISTRY[loss_type]()


def add_loss_tensor(loss_config, output_type, pred_tensor, gt_tensor, mask):
    """The configured loss of `output_type` between prediction and ground truth under `mask`, times its weight."""
    loss_type, loss_weight = get_loss_type_and_weight(loss_config, output_type)
    if loss_type is None:
        return torch.zeros_like(pred_tensor)
    return build_loss(loss_type)(pred_tensor, gt_tensor, weights=mask) * loss_weight
