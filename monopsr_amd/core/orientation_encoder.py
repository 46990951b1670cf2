"""Angle-bin encoding of the observation angle (host side: label preparation, and the decode used for single
values; the batched decode of predictions runs in mpsr_format_boxes).  Function names and return values follow the
numpy half of the reference's core/orientation_encoder.py:6-107.

Bins: `num_bins` equal sectors of the circle, bin 0 centred on 0 rad, counting counter-clockwise.
"""
import math

import numpy as np

TWO_PI = 2.0 * math.pi


def np_wrap_to_pi(angles):
    """Angles -> [-pi, pi) (values exactly at +-pi may land on either end)."""
    return np.mod(np.asarray(angles) + math.pi, TWO_PI) - math.pi


def np_orientation_to_angle_bin(orientation, num_bins, overlap=0.0):
    """-> (bin index, residual to EVERY bin centre (num_bins,), one-hot of the valid bins).

    `overlap` (radians) makes a neighbouring bin valid too when the angle lies that close to the shared boundary:
    the upper neighbour always, the lower neighbour only across the 0 / num_bins-1 wrap -- the reference appends the
    lower neighbour inside its wrap branch (orientation_encoder.py:66-73) and this keeps that behaviour."""
    width = TWO_PI / num_bins
    angle = orientation % TWO_PI
    shifted = (angle + 0.5 * width) % TWO_PI            # bin k now spans [k * width, (k + 1) * width)
    k = int(shifted / width)
    offset_in_bin = shifted - (k + 0.5) * width          # signed distance from the centre of bin k
    delta = angle - width * np.arange(num_bins)
    residuals = np.arctan2(np.sin(delta), np.cos(delta))
    valid = np.zeros(num_bins)
    valid[k] = 1
    if overlap != 0.0:
        if abs(0.5 * width - offset_in_bin) < overlap:
            valid[(k + 1) % num_bins] = 1
        elif abs(-0.5 * width - offset_in_bin) < overlap and k == 0:
            valid[num_bins - 1] = 1
    return k, residuals, valid


def np_angle_bin_to_orientation(angle_bin, residual, num_bins):
    """Bin index + residual from its centre -> orientation, folded once into [-pi, pi]."""
    angle = angle_bin * (TWO_PI / num_bins) + residual
    if angle < -math.pi:
        return angle + TWO_PI
    if angle > math.pi:
        return angle - TWO_PI
    return angle
