"""Name logic of the weight importer (reference: core/checkpoint_utils.py:64-117)."""
import numpy as np
import pytest

from monopsr_amd.core import checkpoint_utils as cu
from monopsr_amd.core import weights as W


def _od_api_checkpoint(seed):
    src = W.synthetic_weights(seed=seed, width_div=8, decoder=False, heads=False)
    return {k.replace("FirstStageFeatureExtractor_crop/", "FirstStageFeatureExtractor/"): v for k, v in src.items()}


def test_od_api_checkpoint_fills_both_trunks(tmp_path):
    model = W.synthetic_weights(seed=1, width_div=8, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    ckpt = _od_api_checkpoint(2)
    ckpt["SecondStageFeatureExtractor/resnet_v1_101/block4/unit_1/bottleneck_v1/conv1/weights"] = np.zeros((1, 1, 4, 4))
    path = str(tmp_path / "od.npz")
    cu.save_npz(path, ckpt)
    restored = cu.restore_obj_detection_api_weights(model, cu.load_npz(path))
    trunk_vars = [k for k in model if k.startswith("FirstStageFeatureExtractor_")]
    assert sorted(restored) == sorted(trunk_vars)
    a = "FirstStageFeatureExtractor_crop/resnet_v1_101/block3/unit_7/bottleneck_v1/conv2/weights"
    b = a.replace("_crop/", "_full/")
    np.testing.assert_array_equal(model[a], model[b])
    np.testing.assert_array_equal(model[a], ckpt[a.replace("_crop/", "/")])
    # decoder / heads keep their initial values
    fresh = W.synthetic_weights(seed=1, width_div=8, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    np.testing.assert_array_equal(model["squash/1x1_conv/weights"], fresh["squash/1x1_conv/weights"])


def test_partial_checkpoint_and_shape_mismatch():
    model = W.synthetic_weights(seed=1, width_div=8)
    ckpt = _od_api_checkpoint(2)
    key = "FirstStageFeatureExtractor/resnet_v1_101/conv1/weights"
    partial = {key: ckpt[key]}
    assert cu.restore_obj_detection_api_weights(model, partial) == [
        "FirstStageFeatureExtractor_crop/resnet_v1_101/conv1/weights"]
    bad = {key: np.zeros((3, 3, 3, 8), np.float32)}
    with pytest.raises(ValueError):
        cu.restore_obj_detection_api_weights(model, bad)
    assert cu.restore_obj_detection_api_weights(model, bad, strict_shapes=False) == []


def test_monopsr_checkpoint_ignores_optimizer_slots():
    model = W.synthetic_weights(seed=3, width_div=8)
    src = W.synthetic_weights(seed=4, width_div=8)
    ckpt = dict(src)
    ckpt["output/lwh/lwh/weights/Adam"] = np.zeros_like(src["output/lwh/lwh/weights"])
    ckpt["global_step"] = np.array(1234)
    restored = cu.restore_monopsr_weights(model, ckpt)
    assert len(restored) == len(src)
    np.testing.assert_array_equal(model["output/alpha/weights"], src["output/alpha/weights"])


def test_expected_variable_table_has_reference_names():
    names = cu.expected_variables()
    for must in ("FirstStageFeatureExtractor_crop/resnet_v1_101/block3/unit_23/bottleneck_v1/conv3/BatchNorm/gamma",
                 "FirstStageFeatureExtractor_full/resnet_v1_101/conv1/weights", "squash/1x1_conv/biases",
                 "map_decoder/conv3/conv3_2/BatchNorm/moving_variance",
                 "output/inst_xyz_map_local/inst_xyz_map_local/weights", "output/proposal_fc/proposal_fc/img_fc/weights",
                 "output/regression_fc/regression_fc/fc1/biases", "output/cen_z_offs/cen_z/weights"):
        assert must in names, must
    assert names["output/proposal_fc/proposal_fc/fc0/weights"] == (1043, 1024)
    assert "map_decoder/conv2/conv2_1/BatchNorm/gamma" not in names  # slim default: no scale in the decoder BN
