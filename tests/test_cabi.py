"""CPU checks of the drop-in boundary: libmonopsr_hip.so builds for gfx950 here (no GPU), loads, and exports every
symbol include/monopsr_hip.h declares; the ctypes binding covers exactly that set.  No compute calls."""
import ctypes
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "monopsr_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mpsr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_path():
    names = _declared()
    for must in ("mpsr_nn_distance_fwd", "mpsr_nn_distance_bwd", "mpsr_approx_match", "mpsr_match_cost",
                 "mpsr_match_cost_grad", "mpsr_crop_and_resize", "mpsr_trunk_fwd", "mpsr_squash_decoder_fwd",
                 "mpsr_heads_fwd", "mpsr_last_error"):
        assert must in names


def test_library_builds_and_exports_every_declared_symbol():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "monopsr_amd", "csrc")], stdout=subprocess.DEVNULL)
    so = os.path.join(ROOT, "monopsr_amd", "libmonopsr_hip.so")
    assert os.path.exists(so)
    out = subprocess.check_output(["nm", "-D", "--defined-only", so]).decode()
    exported = set(re.findall(r" T (mpsr_[a-z0-9_]+)", out))
    missing = [n for n in _declared() if n not in exported]
    assert not missing, missing


def test_ctypes_binding_matches_header_and_loads():
    from monopsr_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared()
    lib = _lib.lib()  # dlopen + every symbol bound; works without a GPU
    assert lib.mpsr_abi_version() == _lib.ABI_VERSION
    assert isinstance(lib.mpsr_last_error(), bytes)  # (the message of this thread's most recent failing call, if any)
    # size helpers are pure host functions
    assert lib.mpsr_approx_match_temp_floats(2, 5, 7) == 2 * (5 + 7) * 11
    # ABI 6: the fused loss' scratch for its level-culling form = the state + both clouds re-ordered + their permutations;
    # host semantics and clouds beyond the in-LDS sort's 4096 points have no such form (plain size)
    assert lib.mpsr_emd_loss_temp_floats(2, 5, 7, 0) == 2 * (5 + 7) * 11 + 2 * (5 + 7) * 4
    assert lib.mpsr_emd_loss_temp_floats(2, 5, 7, 1) == lib.mpsr_emd_temp_floats(2, 5, 7, 1)
    assert lib.mpsr_emd_loss_temp_floats(1, 5000, 7, 0) == lib.mpsr_emd_temp_floats(1, 5000, 7, 0)
    assert lib.mpsr_emd_loss_temp_floats(0, 5, 7, 0) == 0
    assert lib.mpsr_trunk_workspace_bytes(0, 48, 48) == 0
    assert lib.mpsr_trunk_workspace_bytes(2, 48, 48) > 0


def test_struct_layouts_match_the_header():
    from monopsr_amd import _lib
    assert ctypes.sizeof(_lib.Layer) == 6 * 4 + 2 * 8
    assert ctypes.sizeof(_lib.HeadConsts) == 5 * 4 + 2 * 4
    assert ctypes.sizeof(_lib.HeadOutputs) == 11 * ctypes.sizeof(ctypes.c_void_p)
    # (int32 + padding before the event handle; tags pointer; ABI 5: two int32 options)
    assert ctypes.sizeof(_lib.NetOpts) == 8 + 8 + 8 + 8 + 8 + 4 + 4
    assert ctypes.sizeof(_lib.ConvOpts) == 8
    assert ctypes.sizeof(_lib.PackJob) == 8 + 8 + 4 * 4 + 8


def test_host_side_argument_checks_need_no_gpu():
    """Shape errors are reported before anything touches the device (status 1 + message)."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    assert lib.mpsr_nn_distance_fwd(-1, 4, None, 4, None, None, None, None, None, None) == 1
    assert b"negative" in lib.mpsr_last_error()
    assert lib.mpsr_nn_distance_fwd(0, 4, None, 4, None, None, None, None, None, None) == 0  # empty batch: no-op
    assert lib.mpsr_nn_distance_fwd(2, 4, None, 0, None, None, None, None, None, None) == 1
    assert lib.mpsr_conv2d_nhwc_f32(None, 1, 4, 4, 6, None, None, None, None, 8, 1, 1, 1, 0, 1, None, 0, None) == 1
    assert b"multiple of 4" in lib.mpsr_last_error()
    assert lib.mpsr_im2col_root(None, 1, 48, 48, None, 150, None) == 1


def test_relu_bitmask_family_host_side():
    """The training-side entry points added in round 4: size helper and shape predicates are pure host functions, and
    argument errors are reported before anything touches the device."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    assert lib.mpsr_relu_bitmask_words(36864, 1024) == 1152 * 1024
    assert lib.mpsr_relu_bitmask_words(33, 8) == 2 * 8 and lib.mpsr_relu_bitmask_words(0, 8) == 0
    assert lib.mpsr_conv1x1_masked_applies(36864, 256, 1024) == 1   # block3 conv1's data gradient
    assert lib.mpsr_conv1x1_masked_applies(36864, 1024, 256) == 1   # block3 conv3's
    assert lib.mpsr_conv1x1_masked_applies(36864, 128, 512) == 0    # short K: conv + relu_grad
    assert lib.mpsr_conv1x1_masked_applies(36864, 256, 24) == 0     # N not a multiple of 32
    assert lib.mpsr_relu_bitmask(None, 64, 6, None, None) == 1 and b"multiple of 4" in lib.mpsr_last_error()
    assert lib.mpsr_relu_bitmask(None, 0, 8, None, None) == 0       # empty: no-op
    assert lib.mpsr_conv1x1_masked_f32(None, 64, 256, None, None, None, None, None, 64, None) == 1
    assert b"null" in lib.mpsr_last_error()
    assert lib.mpsr_conv1x1_relu_bitmask_f32(None, 0, 256, None, None, None, 1, None, None, 64, None) == 0
    assert lib.mpsr_conv2d_relu_masked_f32(None, 1, 12, 12, 64, None, None, None, 64, 3, 3, 4, None, 0, None) == 1
    assert lib.mpsr_adam_step_lr_dev(None, None, None, None, 0, None, 0.9, 0.999, 1e-8, 1.0, None) == 0
    assert lib.mpsr_adam_step_lr_dev(None, None, None, None, 4, None, 0.9, 0.999, 1e-8, 1.0, None) == 1


@pytest.mark.parametrize("B,H,C,N,dil", [(64, 12, 128, 128, 2), (3, 12, 64, 64, 1), (2, 24, 64, 64, 1),
                                         (8, 12, 256, 256, 4), (2, 48, 64, 64, 1)])
def test_conv2d_plan_counts_the_tiles_the_f3x3_launch_runs(B, H, C, N, dil):
    """mpsr_conv2d_plan, kind 4: the F(3x3,3x3) kernels run th x th tiles of 3x3 per pixel sub-grid (th = H / (3 dil):
    1 for block3's crop maps, 2 for block2, 4 for block1, 8 / 16 for the small-batch decoder) -- the plan's executed
    multiply-adds are 25 per tile and channel pair for exactly that tile count (one helper shared with the launcher)."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    kind, ex = ctypes.c_int(-1), ctypes.c_double(0)
    lib.mpsr_debug_set_conv_winograd(3)
    try:
        _lib.check(lib.mpsr_conv2d_plan(B, H, H, C, N, 3, 3, dil, ctypes.byref(kind), ctypes.byref(ex)))
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    th = H // (3 * dil)
    assert kind.value == 4
    # 25 products per F(3x3,3x3) tile; a sub-grid that is ONE tile (th = 1) runs the sixteen-product form (winograd3z.hip)
    assert ex.value == 2.0 * B * dil * dil * th * th * (16 if th == 1 else 25) * C * N
