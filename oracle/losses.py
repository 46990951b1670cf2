"""CPU restatement (numpy, float64 accumulation) of the losses MonoPSR's training graph sums
(monopsr_model.py:554-958 through builders/loss_builder.py:59-85).

TEST INFRASTRUCTURE ONLY: imported by tests/; never by the product path.

Pinning: the smooth-L1 and softmax classes are vendored OD-API code whose own tests hold known answers
(object_detection/core/losses_test.py:87-107 -> 7.695; :490-544 -> -1.5*log(.5) and the anchor-wise matrix);
tests/test_oracle_geometry.py re-expresses them; the sigmoid focal loss (losses.py:223-280) is pinned by the nine
known answers of losses_test.py:223-487.  tf.losses.huber_loss and the SUM_BY_NONZERO_WEIGHTS reduction live in
TensorFlow 1.8 (not vendored): restated from its documented definition.  BerHu / sigmoid-CE / the model's loss
wiring have no reference test: PARITY UNPINNED for those.
"""
import numpy as np


def huber(pred, target, delta=1.0):
    """tf.losses.huber_loss elementwise term: 0.5 q^2 + delta (|e| - q), q = min(|e|, delta)."""
    e = np.abs(np.asarray(pred, np.float64) - np.asarray(target, np.float64))
    q = np.minimum(e, delta)
    return 0.5 * q * q + delta * (e - q)


def weighted_smooth_l1(pred, target, weights, delta=1.0):
    """object_detection/core/losses.py:118-157 WeightedSmoothL1LocalizationLoss: huber * weights (broadcast),
    Reduction.NONE, summed over axis 2.  weights: (batch, anchors) or (batch, anchors, 1)."""
    w = np.asarray(weights, np.float64)
    if w.ndim == 2:
        w = w[..., None]
    return (huber(pred, target, delta) * w).sum(2)


def weighted_nonzero_smooth_l1(pred, target, weights, delta=1.0):
    """losses_custom.py:93-132 WeightedNonZeroSmoothL1LocalizationLoss: Reduction.SUM_BY_NONZERO_WEIGHTS =
    sum(huber * w) / (number of non-zero entries of w broadcast to the loss shape); 0 when there are none."""
    l = huber(pred, target, delta)
    w = np.broadcast_to(np.asarray(weights, np.float64), l.shape)
    n = np.count_nonzero(w)
    return float((l * w).sum() / n) if n else 0.0


def weighted_softmax(logits, target, weights, logit_scale=1.0):
    """object_detection/core/losses.py:283-317: softmax cross-entropy per row, reshaped to weights' shape, * weights."""
    z = np.asarray(logits, np.float64) / logit_scale
    t = np.asarray(target, np.float64)
    c = z.shape[-1]
    z2, t2 = z.reshape(-1, c), t.reshape(-1, c)
    m = z2.max(1, keepdims=True)
    lse = m + np.log(np.exp(z2 - m).sum(1, keepdims=True))
    ce = (t2 * (lse - z2)).sum(1)
    w = np.asarray(weights, np.float64)
    return ce.reshape(w.shape) * w


def sigmoid_ce(logits, target):
    """losses_custom.py:201-232 SigmoidClassificationLoss: tf.nn.sigmoid_cross_entropy_with_logits, unweighted."""
    x, z = np.asarray(logits, np.float64), np.asarray(target, np.float64)
    return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))


def sigmoid_focal(logits, target, weights, gamma=2.0, alpha=0.25):
    """object_detection/core/losses.py:223-280 SigmoidFocalClassificationLoss: sigmoid cross-entropy per entry times
    (1 - p_t)^gamma and the alpha balance, times weights (batch, anchors) broadcast over classes.  gamma = 0 / None
    and alpha = None switch the factors off as in the reference (`if self._gamma`, `if self._alpha is not None`)."""
    x, z = np.asarray(logits, np.float64), np.asarray(target, np.float64)
    ce = sigmoid_ce(x, z)
    p = 1.0 / (1.0 + np.exp(-x))
    p_t = z * p + (1 - z) * (1 - p)
    mod = np.power(1.0 - p_t, gamma) if gamma else 1.0
    bal = (z * alpha + (1 - z) * (1 - alpha)) if alpha is not None else 1.0
    return mod * bal * ce * np.asarray(weights, np.float64)[..., None]


def weighted_berhu(pred, target, weights):
    """losses_custom.py:45-90 WeightedBerHu."""
    e = np.asarray(pred, np.float64) - np.asarray(target, np.float64)
    a = np.abs(e)
    th = a.max() / 5.0
    per = np.where(a <= th, a, (e * e + th * th) / (2 * th))
    w = np.asarray(weights, np.float64)
    n = np.count_nonzero(w)
    return float((per * w).sum() / n) if n > 0 else 0.0


def model_loss(out, gt, num_boxes, loss_config, num_alpha_bins=12):
    """monopsr_model.py:554-958 for model 000's output set (alpha 'dc', lwh/cen_y/cen_z 'offset', view_ang 'est',
    inst_xyz_map_global 'projection', inst_depth_map_global 'map').  `out` / `gt` are dicts of numpy arrays keyed
    as in core/constants.py; gt also holds 'alpha_valid_bins'.  loss_config: name -> [type, weight, ...].
    Returns (losses_dict, total)."""
    ones = np.ones((1, num_boxes, 1))
    d, total = {}, 0.0

    def wt(name):
        return float(loss_config[name][1])

    m = gt["valid_mask_maps"]
    v = weighted_nonzero_smooth_l1(out["inst_xyz_map_local"], gt["inst_xyz_map_local"], m) \
        * wt("inst_xyz_map_local") / num_boxes
    d["inst_xyz_map_local"] = v
    for key, cfg in (("lwh_offs", "lwh"), ("cen_z_offs", "cen_z"), ("cen_y_offs", "cen_y")):
        if key in out:
            d[key] = (weighted_smooth_l1(out[key][None], gt[key][None], ones) * wt(cfg)).sum() / num_boxes
    eps = float(loss_config["alpha_cls"][2])
    bins = np.asarray(gt["alpha_bins"]).reshape(-1).astype(np.int64)
    one_hot = np.full((num_boxes, num_alpha_bins), eps / num_alpha_bins)
    one_hot[np.arange(num_boxes), bins] = 1.0 - eps
    d["alpha_bins"] = (weighted_softmax(out["alpha_bins"][None], one_hot[None], ones)
                       * wt("alpha_cls")).sum() / num_boxes
    d["alpha_regs"] = (weighted_smooth_l1(out["alpha_regs"][None], gt["alpha_regs"][None],
                                          np.asarray(gt["alpha_valid_bins"])[None]) * wt("alpha_reg")).sum() \
        / num_boxes
    if "proj_err_norm" in out:
        pe = np.asarray(out["proj_err_norm"]).reshape(1, -1, 1)
        d["proj_err"] = weighted_nonzero_smooth_l1(pe, np.zeros_like(pe), ones) * wt("inst_xyz_map_global")
    if "inst_depth_map_global" in out:
        d["inst_depth_map_global"] = weighted_nonzero_smooth_l1(
            out["inst_depth_map_global"], gt["inst_depth_map_global"], m) * wt("inst_depth_map_global") / num_boxes
    for k in d:
        total += d[k]
    return d, total
