"""Point-cloud losses of the reference (core/losses_custom.py:135-198), over the HIP ops."""
from monopsr_amd.tf_ops.approxmatch import tf_approxmatch
from monopsr_amd.tf_ops.nn_distance import tf_nndistance


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
        match = tf_approxmatch.approx_match(pred, tgt)
        distances = tf_approxmatch.match_cost(pred, tgt, match)
        return distances.sum() / float(batch_size)


class ChamferDistance:
    """Computes the chamfer distance between two point clouds."""

    def __call__(self, prediction_tensor, target_tensor, weights):
        batch_size, pred, tgt = _masked_points(prediction_tensor, target_tensor, weights)
        dist1, _, dist2, _ = tf_nndistance.nn_distance(pred, tgt)
        return (dist1.sum() + dist2.sum()) / float(batch_size)
