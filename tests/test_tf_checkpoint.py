"""TensorFlow V2 checkpoint ("tensor bundle") reader / writer: checksum known answers, format constants, round
trips through small and multi-block tables, corruption detection, and the reference's restore name logic driven
from a bundle on disk."""
import os
import struct

import numpy as np
import pytest

from monopsr_amd.core import checkpoint_utils, tf_checkpoint as T


def test_crc32c_known_answers():
    # RFC 3720 B.4 test vectors
    assert T.crc32c(b"123456789") == 0xE3069283
    assert T.crc32c(bytes(32)) == 0x8A9136AA
    assert T.crc32c(bytes([0xFF] * 32)) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"6789", T.crc32c(b"12345")) == 0xE3069283          # incremental
    big = np.random.default_rng(0).integers(0, 256, 100003, dtype=np.uint8).tobytes()
    assert T.crc32c(big[50001:], T.crc32c(big[:50001])) == T.crc32c(big)
    # LevelDB's crc mask: rotate right 15, add 0xa282ead8 (crc32c.h); unmask(mask(x)) == x
    m = T.mask_crc(0x12345678)
    rot = (m - 0xa282ead8) & 0xffffffff
    assert ((rot >> 17) | (rot << 15)) & 0xffffffff == 0x12345678


def _tensors(rng, n, prefix="FirstStageFeatureExtractor/resnet_v1_101/"):
    out = {}
    for i in range(n):
        shape = tuple(int(v) for v in rng.integers(1, 6, rng.integers(0, 5)))
        out["%sblock%d/unit_%d/bottleneck_v1/conv%d/weights" % (prefix, i % 4, i // 4, i % 3)] = \
            rng.standard_normal(shape).astype(np.float32)
    out["global_step"] = np.asarray(142000, np.int64)
    out["some/int32"] = rng.integers(-5, 5, (3, 2)).astype(np.int32)
    out["some/half"] = rng.standard_normal((4,)).astype(np.float16)
    out["some/empty"] = np.zeros((0, 3), np.float32)
    return out


@pytest.mark.parametrize("n,block_size", [(5, 256 << 10), (300, 512), (40, 64)])
def test_round_trip(tmp_path, n, block_size):
    rng = np.random.default_rng(n)
    tensors = _tensors(rng, n)
    prefix = T.write_checkpoint(str(tmp_path / "model.ckpt-1"), tensors, block_size=block_size)
    assert os.path.exists(prefix + ".index") and os.path.exists(prefix + ".data-00000-of-00001")
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xdb4775248b80fb57 and len(raw) >= 48
    got = T.read_checkpoint(prefix)
    assert set(got) == set(tensors)
    for k, v in tensors.items():
        assert got[k].dtype == v.dtype and got[k].shape == v.shape
        np.testing.assert_array_equal(got[k], v)
    listed = dict(T.list_variables(prefix + ".index"))
    assert listed["global_step"] == [] and listed["some/empty"] == [0, 3]
    one = T.read_checkpoint(prefix + ".data-00000-of-00001", names=["some/int32"])
    assert list(one) == ["some/int32"]


def test_corruption_is_detected(tmp_path):
    tensors = _tensors(np.random.default_rng(1), 20)
    prefix = T.write_checkpoint(str(tmp_path / "m"), tensors)
    data = bytearray(open(prefix + ".data-00000-of-00001", "rb").read())
    data[len(data) // 2] ^= 0x40
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(data))
    with pytest.raises(T.CheckpointError, match="payload checksum"):
        T.read_checkpoint(prefix)
    assert len(T.read_checkpoint(prefix, verify=False)) == len(tensors)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[10] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(T.CheckpointError, match="block checksum"):
        T.read_checkpoint(prefix)
    open(prefix + ".index", "wb").write(b"not a table" * 10)
    with pytest.raises(T.CheckpointError, match="magic"):
        T.read_checkpoint(prefix)
    with pytest.raises(T.CheckpointError):
        T.read_checkpoint(str(tmp_path / "m"), names=["nope"])


def test_snappy_block_decoder():
    # literal "abcd", copy(offset 4, len 4) twice with 1- and 2-byte offsets -> "abcdabcdabcd"
    blob = bytes([12, (4 - 1) << 2]) + b"abcd" + bytes([0b00000001 | ((4 - 4) << 2), 4]) + bytes([((4 - 1) << 2) | 2, 4, 0])
    assert T._snappy_uncompress(blob) == b"abcdabcdabcd"


def test_restore_from_a_bundle_with_the_reference_name_logic(tmp_path):
    """OD-API checkpoint (one trunk under FirstStageFeatureExtractor/) -> both trunks; MonoPSR checkpoint (own
    names + optimizer slots) -> every variable; through files on disk and the `checkpoint` state file."""
    from monopsr_amd.core import weights as W
    model = W.synthetic_weights(seed=0, width_div=8, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    rng = np.random.default_rng(2)
    od = {}
    for name, arr in model.items():
        if name.startswith(checkpoint_utils.TRUNK_SCOPES[0]):
            od[checkpoint_utils.OD_API_SCOPE + name[len(checkpoint_utils.TRUNK_SCOPES[0]):]] = \
                rng.standard_normal(arr.shape).astype(np.float32)
    od["SecondStageBoxPredictor/weights"] = np.zeros((3, 3), np.float32)         # ignored: not ours
    T.write_checkpoint(str(tmp_path / "od" / "model.ckpt"), od)
    T.write_checkpoint_state(str(tmp_path / "od"), "model.ckpt")
    ck = checkpoint_utils.load_checkpoint(str(tmp_path / "od"))                   # directory with a state file
    restored = checkpoint_utils.restore_obj_detection_api_weights(model, ck)
    n_trunk = sum(k.startswith(checkpoint_utils.TRUNK_SCOPES[0]) for k in model)
    assert len(restored) == 2 * n_trunk
    for scope in checkpoint_utils.TRUNK_SCOPES:
        k = scope + "resnet_v1_101/conv1/weights"
        np.testing.assert_array_equal(model[k], od[checkpoint_utils.OD_API_SCOPE + "resnet_v1_101/conv1/weights"])
    # a MonoPSR checkpoint written by save_checkpoint and read back
    prefix = checkpoint_utils.save_checkpoint(str(tmp_path / "run" / "monopsr"), dict(model, **{
        "output/lwh/lwh/weights/Adam": np.ones((2, 2), np.float32)}), global_step=2000)
    assert prefix.endswith("monopsr-00002000")
    fresh = W.synthetic_weights(seed=5, width_div=8, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    got = checkpoint_utils.restore_monopsr_weights(fresh, checkpoint_utils.load_checkpoint(str(tmp_path / "run")))
    assert set(got) == set(model)
    for k in model:
        np.testing.assert_array_equal(fresh[k], model[k])


def test_index_keys_are_short_separators():
    """Index entries follow the table builder's rule (shortest separator between blocks, short successor for the
    last), and a multi-block file written that way reads back."""
    import numpy as np
    from monopsr_amd.core import tf_checkpoint as T
    rng = np.random.default_rng(5)
    for _ in range(2000):
        a = bytes(rng.integers(97, 100, rng.integers(1, 6)).astype(np.uint8))
        b = bytes(rng.integers(97, 100, rng.integers(1, 6)).astype(np.uint8))
        lo, hi = sorted([a, b])
        if lo == hi:
            continue
        sep = T._shortest_separator(lo, hi)
        assert lo <= sep < hi and len(sep) <= len(lo)
        assert T._short_successor(lo) >= lo
    assert T._shortest_separator(b"block1/unit_1", b"block3/unit_2") == b"block2"
    assert T._short_successor(b"\xff\xffa") == b"\xff\xffb"


def test_many_small_blocks_round_trip(tmp_path):
    import numpy as np
    from monopsr_amd.core import tf_checkpoint as T
    tensors = {"scope/layer_%03d/weights" % i: np.full((3,), i, np.float32) for i in range(200)}
    T.write_checkpoint(str(tmp_path / "m"), tensors, block_size=256)  # dozens of data blocks
    back = T.read_checkpoint(str(tmp_path / "m"))
    assert sorted(back) == sorted(tensors)
    for k, v in tensors.items():
        np.testing.assert_array_equal(back[k], v)
