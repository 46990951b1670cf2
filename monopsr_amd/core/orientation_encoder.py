"""Angle-bin encoding of the observation angle, mirroring the numpy half of core/orientation_encoder.py:6-107 of
the reference (host-side label preparation and the decode used when formatting predictions; the device-side decode
lives in mpsr_format_boxes)."""
import numpy as np


def np_wrap_to_pi(angles):
    """Wrap angles between [-pi, pi]. Angles right at -pi or pi may flip."""
    return (angles + np.pi) % (2 * np.pi) - np.pi


def np_orientation_to_angle_bin(orientation, num_bins, overlap=0.0):
    """-> (angle_bin, residuals to every bin centre (num_bins,), one_hot_valid_bins).  Bin 0 is centred on 0 rad.
    With overlap the upper neighbour becomes valid within `overlap` of the upper boundary; the lower neighbour only
    when it wraps around to the last bin, as orientation_encoder.py:66-73 has it."""
    two_pi = 2 * np.pi
    orientation_wrapped = orientation % two_pi
    angle_per_bin = two_pi / num_bins
    shifted_angle = (orientation_wrapped + angle_per_bin / 2) % two_pi
    best_angle_bin = int(shifted_angle / angle_per_bin)
    best_residual = shifted_angle - (best_angle_bin * angle_per_bin + angle_per_bin / 2)
    bin_centres = angle_per_bin * np.arange(num_bins)
    residuals = np.arctan2(np.sin(orientation_wrapped - bin_centres), np.cos(orientation_wrapped - bin_centres))
    valid_bins = [best_angle_bin]
    if overlap != 0.0:
        bin_centre = best_angle_bin * angle_per_bin
        actual_angle = bin_centre + best_residual
        if np.abs(bin_centre + 0.5 * angle_per_bin - actual_angle) < overlap:
            valid_bins.append((best_angle_bin + 1) % num_bins)
        elif np.abs(bin_centre - 0.5 * angle_per_bin - actual_angle) < overlap:
            if best_angle_bin - 1 < 0:
                valid_bins.append(num_bins - 1)
    one_hot_valid_bins = np.zeros(num_bins)
    one_hot_valid_bins[np.asarray(valid_bins)] = 1
    return best_angle_bin, residuals, one_hot_valid_bins


def np_angle_bin_to_orientation(angle_bin, residual, num_bins):
    """Bin index + residual from the bin centre -> orientation in [-pi, pi]."""
    two_pi = 2 * np.pi
    angle = angle_bin * (two_pi / num_bins) + residual
    if angle < -np.pi:
        angle = angle + two_pi
    if angle > np.pi:
        angle = angle - two_pi
    return angle
