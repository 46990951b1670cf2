"""Loss selection by config, mirroring the reference's builders/loss_builder.py:8-85."""
import torch

from monopsr_amd.core import losses, losses_custom


def get_loss_type_and_weight(loss_config, output_rep):
    if not hasattr(loss_config, output_rep):
        raise ValueError('Loss not configured for output_rep:', output_rep)
    this_loss_config = getattr(loss_config, output_rep)
    return this_loss_config[0], this_loss_config[1]


def build_loss(loss_type):
    if loss_type == 'berHu':
        return losses_custom.WeightedBerHu()
    elif loss_type == 'chamfer_dist':
        return losses_custom.ChamferDistance()
    elif loss_type == 'emd':
        return losses_custom.EarthMoversDistance()
    elif loss_type == 'smooth_l1':
        return losses.WeightedSmoothL1LocalizationLoss()
    elif loss_type == 'smooth_l1_nonzero':
        return losses_custom.WeightedNonZeroSmoothL1LocalizationLoss()
    elif loss_type == 'softmax':
        return losses.WeightedSoftmaxClassificationLoss()
    elif loss_type == 'softmax_temp':
        return losses.WeightedSoftmaxClassificationLoss(0.5)
    elif loss_type == 'sigmoid_ce':
        return losses_custom.SigmoidClassificationLoss()
    elif loss_type == 'focal':
        raise ValueError('focal loss is not selected by any MonoPSR config and is not built here', loss_type)
    else:
        raise ValueError('Invalid loss type', loss_type)


def add_loss_tensor(loss_config, output_type, pred_tensor, gt_tensor, mask):
    """loss_builder.py:59-85: the configured loss of `output_type`, times its configured weight."""
    loss_type, loss_weight = get_loss_type_and_weight(loss_config, output_type)
    if loss_type is None:
        return torch.zeros_like(pred_tensor)
    loss_obj = build_loss(loss_type)
    return loss_obj(pred_tensor, gt_tensor, weights=mask) * loss_weight
