"""Parity of the HIP network path (image ops, fp32-MFMA conv/FC, trunk, squash+decoder, heads; all through the
C ABI) against the CPU restatement oracle/net.py on the same seeded inputs and weights.

Tolerance: BASELINE north_star allows 1e-3 relative on float tensors; the fp32-MFMA path is held to 1e-4 of the
tensor's scale here (measured drift is ~1e-6..1e-5; it differs from the oracle only by summation order and by
folding BatchNorm into the weights).
"""
import ctypes

import numpy as np
import pytest
import torch

from oracle import net as onet

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, ref, tol=1e-4, name=""):
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref).max() / scale
    assert err <= tol, "%s: max err / scale = %.3e (scale %.3e)" % (name, err, scale)
    return err


# ------------------------------------------------------------------------------------------- image operators

@pytest.mark.parametrize("C", [3, 8, 64])
def test_crop_and_resize_vs_oracle(C):
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(C)
    img = rng.standard_normal((2, 37, 53, C)).astype(np.float32)
    boxes = np.array([[0.1, 0.2, 0.6, 0.9], [0.0, 0.0, 1.0, 1.0], [-0.2, 0.3, 0.5, 1.3], [0.4, 0.4, 0.4, 0.4],
                      [0.9, 0.1, 0.2, 0.8], [0.25, 0.25, 0.75, 0.5], [2.0, 2.0, 3.0, 3.0]], np.float32)
    ind = np.array([0, 1, 0, 1, 0, 1, 0], np.int32)
    for size in ((48, 48), (24, 24), (5, 7), (1, 1)):
        ref = onet.tf_crop_and_resize(torch.from_numpy(img), boxes, ind, size[0], size[1], 0.0)
        got = dn.crop_and_resize(_dev(img), _dev(boxes), _dev(ind), size, 0.0)
        _close(got, ref, 1e-6, "crop_and_resize %s" % (size,))


@pytest.mark.parametrize("shape,out,ac", [((2, 12, 12, 16), (24, 24), True), ((1, 24, 24, 8), (48, 48), True),
                                          ((1, 37, 124, 3), (32, 122), False), ((2, 10, 7, 4), (10, 7), True),
                                          ((1, 32, 122, 3), (16, 61), True), ((1, 5, 5, 3), (1, 1), True)])
def test_resize_bilinear_vs_oracle(shape, out, ac):
    from monopsr_amd.core import device_net as dn
    x = np.random.default_rng(1).standard_normal(shape).astype(np.float32)
    ref = onet.tf_resize_bilinear(torch.from_numpy(x), out[0], out[1], ac)
    got = dn.resize_bilinear(_dev(x), out, ac)
    _close(got, ref, 1e-6, "resize")


@pytest.mark.parametrize("shape,k,s,pad", [((2, 24, 24, 64), 3, 2, "SAME"), ((2, 12, 12, 32), 2, 2, "VALID"),
                                           ((1, 80, 304, 8), 3, 2, "SAME"), ((1, 7, 9, 3), 3, 2, "SAME"),
                                           ((1, 24, 24, 5), 2, 2, "VALID")])
def test_max_pool_vs_oracle(shape, k, s, pad):
    from monopsr_amd.core import device_net as dn
    x = np.random.default_rng(2).standard_normal(shape).astype(np.float32)
    ref = onet.tf_max_pool(torch.from_numpy(x), k, s, pad)
    got = dn.max_pool(_dev(x), k, s, pad)
    np.testing.assert_array_equal(got.cpu().numpy(), ref.numpy())


# ------------------------------------------------------------------------------------------- conv / FC kernel

def _conv_ref(x, w_hwio, bias, residual, rate, relu):
    y = onet.tf_conv2d(torch.from_numpy(x).double(), torch.from_numpy(w_hwio).double(), rate=rate)
    if bias is not None:
        y = y + torch.from_numpy(bias).double()
    if residual is not None:
        y = y + torch.from_numpy(residual).double()
    return torch.relu(y) if relu else y


CONV_CASES = [
    # B, H, W, C, N, k, rate, bias, residual, relu
    (2, 12, 12, 64, 64, 1, 1, True, False, True),
    (2, 12, 12, 64, 256, 1, 1, True, True, True),
    (3, 12, 12, 64, 64, 3, 1, True, False, True),
    (2, 12, 12, 128, 128, 3, 2, True, False, True),
    (2, 12, 12, 256, 256, 3, 4, True, False, True),
    (1, 24, 24, 32, 48, 3, 1, True, False, True),
    (1, 9, 11, 36, 40, 3, 2, False, True, False),     # ragged M, N, C (C % 32 != 0)
    (5, 1, 1, 96, 27, 1, 1, True, False, False),      # FC with a narrow N
    (7, 3, 5, 8, 3, 3, 1, True, False, False),        # xyz-like: N = 3
    (1, 48, 48, 16, 3, 3, 1, True, False, False),
    # N <= 4, C % 32 == 0, 3x3 d1: the direct narrow-conv kernel when no tile is forced (tile -1), MFMA otherwise
    (2, 20, 20, 64, 3, 3, 1, True, False, False),
    (1, 48, 48, 128, 3, 3, 1, True, False, False),
    (2, 9, 11, 32, 4, 3, 1, False, False, False),
    (1, 17, 16, 96, 1, 3, 1, True, False, False),
    (3, 33, 18, 64, 2, 3, 1, True, False, False),
]


@pytest.mark.parametrize("depth", [1, 2])
@pytest.mark.parametrize("classes", [-1, 0, 1])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_vs_fp64(case, tile, classes, depth):
    """Every tile instantiation x border-class tiling off / forced on (also at dilation 1) / heuristic x one or two
    staging register sets."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    B, H, Wd, C, N, k, rate, has_bias, has_res, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((k, k, C, N)) / np.sqrt(k * k * C)).astype(np.float32)  # asymmetric, dense
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
    ref = _conv_ref(x, w, bias, res, rate, relu)
    w_ok, _ = W.fold_conv(w)
    _lib.lib().mpsr_debug_set_conv_tile(tile)
    _lib.lib().mpsr_debug_set_conv_classes(classes)
    _lib.lib().mpsr_debug_set_conv_depth(depth)
    try:
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, _dev(res) if has_res else None, k, k,
                        rate, relu)
    finally:
        _lib.lib().mpsr_debug_set_conv_tile(-1)
        _lib.lib().mpsr_debug_set_conv_classes(-1)
        _lib.lib().mpsr_debug_set_conv_depth(-1)
    _close(got, ref, 2e-6, "conv %s tile %d classes %d depth %d" % (case, tile, classes, depth))


@pytest.mark.parametrize("groups", [0, -8, -40, 3])
@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_stream_k_vs_fp64(case, tile, groups):
    """The stream-K schedule for every tile instantiation and several workgroup counts: the device's own (0), 8 and
    40 workgroups (many tiles per workgroup) and 3 per CU."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    B, H, Wd, C, N, k, rate, has_bias, has_res, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((k, k, C, N)) / np.sqrt(k * k * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
    ref = _conv_ref(x, w, bias, res, rate, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_tile(tile)
    lib.mpsr_debug_set_conv_sched(1, groups)
    try:
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, _dev(res) if has_res else None, k, k,
                        rate, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_tile(-1)
        lib.mpsr_debug_set_conv_sched(-1, 0)
    _close(got, ref, 2e-6, "stream-K conv %s tile %d groups %d" % (case, tile, groups))


@pytest.mark.parametrize("shape", [(64, 12, 256, 256, 1, 1), (64, 12, 256, 256, 3, 4), (96, 12, 128, 512, 1, 1),
                                   (8, 48, 64, 128, 3, 1), (37, 1, 4608, 200, 1, 1)])
def test_stream_k_against_one_tile_per_workgroup(shape):
    """Thousands of units over every XCD: stream-K equals the one-tile-per-workgroup kernel to fp32 summation order
    (bit-identical where no tile is split), is deterministic run to run, and leaves its counters clean (a second
    call on the same scratch gives the same bits)."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    B, H, C, N, k, d = shape
    g = torch.Generator(device="cuda").manual_seed(B * 131 + C)
    x = torch.randn((B, H, H, C), device="cuda", generator=g)
    w = torch.randn((N, k * k * C), device="cuda", generator=g) / np.sqrt(k * k * C)
    bias = torch.randn((N,), device="cuda", generator=g)
    res = torch.randn((B, H, H, N), device="cuda", generator=g)
    nws = lib.mpsr_conv2d_scratch_floats(B, H, H, N)
    ws = torch.full((nws,), float("nan"), device="cuda")  # poisoned scratch: counters must be reset by the library

    def run(split, sched, groups=0):
        y = torch.empty((B, H, H, N), device="cuda")
        lib.mpsr_debug_set_conv_sched(1, groups)
        try:
            _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, H, C, w.data_ptr(), bias.data_ptr(),
                                                res.data_ptr(), y.data_ptr(), N, k, k, d, 1, split, ws.data_ptr(), nws,
                                                _lib.stream()))
        finally:
            lib.mpsr_debug_set_conv_sched(-1, 0)
        return y
    base = run(1, 0)
    for groups in (0, 5, -1000):
        a = run(0, 1, groups)
        b = run(0, 1, groups)
        assert torch.equal(a, b), "stream-K not deterministic (groups %d)" % groups
        err = float((a - base).abs().max() / base.abs().max())
        assert err < 2e-6, "groups %d: %.3e" % (groups, err)


@pytest.mark.parametrize("B,H,Wd,C,N,has_bias,relu", [
    (2, 24, 24, 64, 64, True, True), (1, 48, 48, 32, 128, True, True), (3, 10, 6, 16, 40, False, False),
    (1, 2, 2, 16, 4, True, False), (5, 12, 12, 48, 200, True, True), (4, 48, 48, 256, 128, True, True),
    (2, 24, 24, 512, 256, True, True), (7, 4, 8, 16, 65, True, True)])
@pytest.mark.parametrize("waves", [8, 4])
def test_conv2d_winograd_vs_fp64(B, H, Wd, C, N, has_bias, relu, waves):
    """The Winograd F(2x2,3x3) kernels (3x3, stride 1, SAME, even maps) -- the two-waves-per-SIMD one the library
    uses and the one-wave-per-SIMD one it keeps: ragged tile / channel counts, borders, the decoder's real layer
    shapes.  Tolerance 1e-5 of the tensor scale (the transforms add a few roundings to the direct path's 2e-6)."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B * 1000 + C + N)
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    ref = _conv_ref(x, w, bias, None, 1, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_winograd(1)
    lib.mpsr_debug_set_wino_waves(waves)
    try:
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, 1, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
        lib.mpsr_debug_set_wino_waves(8)
    _close(got, ref, 1e-5, "winograd %s, %d waves" % ((B, H, Wd, C, N), waves))
    lib.mpsr_debug_set_conv_winograd(0)
    try:
        direct = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, 1, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    _close(got, direct, 1e-5, "winograd vs direct")
    assert not torch.equal(got, direct) or C * N < 1024  # it really is a different evaluation


@pytest.mark.parametrize("B,H,Wd,C,N,dil,has_bias,relu", [
    (1, 40, 152, 256, 256, 4, True, True),    # the full-image trunk's block3 conv2 (40 x 152 at dilation 4: 10 x 38 sub-grids)
    (1, 40, 152, 128, 128, 2, True, True),    # ... block2 conv2 (dilation 2: 20 x 76 sub-grids)
    (3, 8, 12, 16, 40, 2, False, False), (2, 12, 6, 32, 65, 3, True, True), (2, 16, 8, 48, 24, 4, True, False),
    (5, 4, 4, 16, 8, 2, True, True)])
@pytest.mark.parametrize("waves", [8, 4])
def test_conv2d_winograd_on_atrous_sub_grids_vs_fp64(B, H, Wd, C, N, dil, has_bias, relu, waves):
    """F(2x2,3x3) on the pixel sub-grids of an atrous 3x3 layer (csrc/winograd.hip with WinoParams::dil: the d^2
    sub-grids of H / d x W / d pixels are independent dense images on strided views): the full-image trunk's block2 /
    block3 layers (reference graph resnet_v1.py:116-127 at output stride 4 on a 160 x 608 input), ragged tile and
    channel counts, several dilations.  <= 1e-5 of the tensor scale against float64 and against the border-class
    implicit GEMM, and really another evaluation."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B * 1000 + C + N + dil)
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    ref = _conv_ref(x, w, bias, None, dil, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_winograd(1)
    lib.mpsr_debug_set_wino_waves(waves)
    try:
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
        lib.mpsr_debug_set_wino_waves(8)
    _close(got, ref, 1e-5, "atrous winograd F(2x2) %s, %d waves" % ((B, H, Wd, C, N, dil), waves))
    lib.mpsr_debug_set_conv_winograd(0)
    try:
        direct = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    _close(got, direct, 1e-5, "atrous winograd vs implicit GEMM")
    assert not torch.equal(got, direct) or C * N < 1024
    # the plan reports the kernel the call takes and the products it issues
    kind, ex = ctypes.c_int(-1), ctypes.c_double(0)
    lib.mpsr_debug_set_conv_winograd(1)
    try:
        _lib.check(lib.mpsr_conv2d_plan(B, H, Wd, C, N, 3, 3, dil, ctypes.byref(kind), ctypes.byref(ex)))
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    assert kind.value == 1 and ex.value == 2.0 * B * (H // 2) * (Wd // 2) * 16 * C * N


UPCONV_CASES = [
    # B, h, w, C, OH, OW, N, align_corners, bias, relu
    (2, 12, 12, 128, 24, 24, 128, True, True, True),    # the decoder's 2x align-corners upsampling, narrow
    (3, 6, 6, 192, 12, 12, 256, True, True, False),     # two GEMM parts, C not a power of two
    (2, 5, 7, 128, 11, 16, 128, True, False, True),     # uneven scales, odd sizes, bands with a ragged tail
    (2, 5, 7, 128, 11, 16, 128, False, True, True),     # legacy scale in / out
    (1, 12, 12, 512, 24, 24, 256, True, True, True),    # conv2_1's real layer shape
    (1, 24, 24, 256, 48, 48, 128, True, True, True),    # conv3_1's real layer shape
    (2, 8, 8, 128, 12, 12, 128, True, True, True),      # 1.5x
]


@pytest.mark.parametrize("B,h,w,C,OH,OW,N,align,has_bias,relu", UPCONV_CASES)
def test_conv3x3_upsampled_vs_fp64(B, h, w, C, OH, OW, N, align, has_bias, relu):
    """tf.image.resize_bilinear -> 3x3 SAME conv as ONE operator (csrc/upconv.hip: tap GEMM on the source map + gather;
    reference net_builder.py:72-77, :81-85) against the two TF-1.8 operators of oracle/net.py in float64: <= 1e-5 of
    the tensor scale (plain fp32 GEMM error; F(4x4,3x3) on the upsampled map is held to 1e-4), deterministic."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B * 1000 + C + N + OH)
    x = rng.standard_normal((B, h, w, C)).astype(np.float32)
    wgt = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    up = onet.tf_resize_bilinear(torch.from_numpy(x).double(), OH, OW, align)
    ref = onet.tf_conv2d(up, torch.from_numpy(wgt).double())
    if has_bias:
        ref = ref + torch.from_numpy(bias).double()
    if relu:
        ref = torch.relu(ref)
    w_ok, _ = W.fold_conv(wgt)
    args = (_dev(x), (OH, OW), _dev(w_ok), _dev(bias) if has_bias else None, relu, align)
    got = dn.conv3x3_upsampled(*args)
    again = dn.conv3x3_upsampled(*args)
    _close(got, ref, 1e-5, "upsampled 3x3 conv %s" % ((B, h, w, C, OH, OW, N),))
    assert torch.equal(got, again), "not deterministic"
    # and against the library's own two-operator chain in fp32
    two = dn.conv2d(dn.resize_bilinear(_dev(x), (OH, OW), align), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3,
                    1, relu)
    _close(got, two, 1e-5, "upsampled 3x3 conv vs resize + conv2d")


def test_conv3x3_upsampled_refuses_what_it_does_not_take():
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    x = torch.zeros((1, 6, 6, 128), device="cuda")
    with pytest.raises(_lib.MpsrError):
        dn.conv3x3_upsampled(x, (12, 12), torch.zeros((64, 9 * 128), device="cuda"))  # N not a multiple of 128


@pytest.mark.parametrize("B,H,Wd,C,N,has_bias,relu", [
    (2, 24, 24, 64, 64, True, True), (1, 48, 48, 32, 128, True, True), (3, 12, 8, 16, 40, False, False),
    (1, 4, 4, 16, 4, True, False), (5, 12, 12, 48, 200, True, True), (4, 48, 48, 256, 128, True, True),
    (2, 24, 24, 512, 256, True, True), (7, 4, 8, 16, 65, True, True), (9, 8, 4, 32, 64, True, False)])
def test_conv2d_winograd4_vs_fp64(B, H, Wd, C, N, has_bias, relu):
    """The Winograd F(4x4,3x3) kernel (csrc/winograd4.hip: 3x3, stride 1, SAME, maps that divide into 4x4 blocks):
    ragged tile / channel counts (tiles not a multiple of 32, N not a multiple of 64), every border case of a 6x6
    patch, the decoder's real layer shapes.  Tolerance 1e-4 of the tensor scale against float64 (measured ~1.5e-5:
    transform constants up to 8 and 1/24; the north star's budget is 1e-3)."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B * 1000 + C + N)
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    ref = _conv_ref(x, w, bias, None, 1, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_winograd(2)
    try:
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, 1, relu, split_k=0)
        again = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, 1, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    _close(got, ref, 1e-4, "winograd F(4x4) %s" % ((B, H, Wd, C, N),))
    assert torch.equal(got, again), "not deterministic"
    lib.mpsr_debug_set_conv_winograd(0)
    try:
        direct = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, 1, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    assert not torch.equal(got, direct) or C * N < 1024  # it really is a different evaluation


@pytest.mark.parametrize("B,H,Wd,C,N,has_bias,relu", [
    (1, 48, 48, 32, 128, True, True), (4, 48, 48, 256, 128, True, True), (2, 24, 24, 512, 256, True, True),
    (3, 12, 8, 16, 128, False, False), (1, 4, 4, 16, 128, True, False), (5, 12, 20, 48, 384, True, True),
    (37, 24, 24, 64, 128, True, True)])
def test_conv2d_winograd4_position_split_vs_fp64(B, H, Wd, C, N, has_bias, relu):
    """The position-split F(4x4,3x3) kernel (csrc/winograd4.hip: two workgroups per 32 tiles x 128 channels, each with 18
    of the 36 positions, partial outputs combined through a ticket): taken when N % 128 == 0 and the scratch has room
    for the partials, which this test provides.  Against float64 (1e-4), against the 64-channel-block kernel (same
    products, another order of the final sums: 1e-5), and repeated runs must agree bit for bit whichever half of a
    pair arrives first."""
    from monopsr_amd import _lib
    from monopsr_amd.core import weights as W
    lib = _lib.lib()
    rng = np.random.default_rng(B * 1000 + C + N)
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    ref = _conv_ref(x, w, bias, None, 1, relu)
    w_ok, _ = W.fold_conv(w)
    xd, wd = _dev(x), _dev(w_ok)
    bd = _dev(bias) if has_bias else None
    tiles = B * (H // 4) * (Wd // 4)
    nws = lib.mpsr_conv2d_scratch_floats(B, H, Wd, N) + B * H * Wd * N + ((tiles + 31) // 32) * (N // 128) * 4 + 256
    ws = torch.full((nws,), float("nan"), device="cuda")  # poisoned: tickets / flags must be reset by the library

    def run(split):
        y = torch.empty((B, H, Wd, N), device="cuda")
        lib.mpsr_debug_set_conv_winograd(2)
        lib.mpsr_debug_set_wino4_split(split)
        try:
            _lib.check(lib.mpsr_conv2d_nhwc_f32(xd.data_ptr(), B, H, Wd, C, wd.data_ptr(),
                                                bd.data_ptr() if has_bias else None, None, y.data_ptr(), N, 3, 3, 1,
                                                int(relu), 0, ws.data_ptr(), nws, _lib.stream()))
        finally:
            lib.mpsr_debug_set_conv_winograd(-1)
            lib.mpsr_debug_set_wino4_split(1)
        return y
    got = run(1)
    _close(got, ref, 1e-4, "winograd F(4x4) split %s" % ((B, H, Wd, C, N),))
    whole = run(0)
    _close(got, whole, 1e-5, "split vs 64-channel-block kernel")
    assert not torch.equal(got, whole)  # it really is the other kernel
    for _ in range(4):
        assert torch.equal(run(1), got), "depends on the order of arrival"


@pytest.mark.parametrize("B,H,Wd,C,N,has_bias,has_res,relu", [
    (4, 12, 12, 256, 1024, True, True, True),     # block3 conv3's shape: two whole row groups
    (3, 12, 12, 1024, 256, True, False, True),    # conv1: 1.5 row groups (rows past M), 32 stages
    (5, 7, 9, 128, 160, False, True, False),      # M = 315 (not a multiple of 32), N % 128 != 0, four stages
    (1, 3, 5, 128, 32, True, False, False),       # one partial tile, one live wave
    (16, 12, 12, 512, 384, True, True, True),     # 24 row groups x three column blocks
    (2, 12, 12, 192, 128, True, True, True),      # six stages
    (40, 12, 12, 256, 1024, True, True, True),    # 480 tiles on 512 workgroups ... 
    (100, 12, 12, 256, 512, True, True, False)])  # ... and 600: one or two tiles per workgroup (both accumulator sets)
def test_conv2d_pointwise_vs_fp64(B, H, Wd, C, N, has_bias, has_res, relu):
    """The persistent pointwise kernel of the wide 1x1 layers (csrc/pointwise.hip) against float64: whole and ragged
    row groups, rows past M, column blocks past N, one to several tiles per workgroup, residual / bias / ReLU;
    deterministic; and the same sums as the implicit GEMM up to the order of the additions."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B * 100 + C + N)
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((1, 1, C, N)) / np.sqrt(C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
    ref = _conv_ref(x, w, bias, res, 1, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()

    def run(mode):
        lib.mpsr_debug_set_conv_pointwise(mode)
        try:
            return dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, _dev(res) if has_res else None, 1, 1,
                             1, relu, split_k=0)
        finally:
            lib.mpsr_debug_set_conv_pointwise(-1)
    kind, flops = ctypes.c_int(-1), ctypes.c_double(0)
    lib.mpsr_debug_set_conv_pointwise(1)
    try:
        _lib.check(lib.mpsr_conv2d_plan(B, H, Wd, C, N, 1, 1, 1, ctypes.byref(kind), ctypes.byref(flops)))
    finally:
        lib.mpsr_debug_set_conv_pointwise(-1)
    assert kind.value == 5 and flops.value == 2.0 * B * H * Wd * C * N
    got = run(1)
    _close(got, ref, 1e-5, "pointwise %s" % ((B, H, Wd, C, N),))
    assert torch.equal(run(1), got), "not deterministic"
    _close(got, run(0), 1e-5, "pointwise vs implicit GEMM")


@pytest.mark.parametrize("M,K,N,has_bias,has_res,relu", [
    (256, 1056, 1024, True, False, True),   # the heads' first FC (concat row padded to a multiple of 32)
    (256, 1024, 27, True, False, False),    # narrow output: one column tile, 5 dead columns
    (256, 1024, 2, False, False, False),
    (100, 160, 40, True, True, True),       # rows and columns past the end, residual
    (1, 128, 33, True, False, True)])
def test_fc_few_rows_vs_fp64(M, K, N, has_bias, has_res, relu):
    """Fully-connected layers with few rows (csrc/pointwise.hip fc_rows_kernel: one 32 x 32 tile per workgroup, its four
    waves split K) against float64, and that the plan reports them."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(M + K + N)
    x = rng.standard_normal((M, 1, 1, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    res = rng.standard_normal((M, 1, 1, N)).astype(np.float32) if has_res else None
    ref = x.reshape(M, K).astype(np.float64) @ w.T.astype(np.float64)
    if has_bias:
        ref = ref + bias
    if has_res:
        ref = ref + res.reshape(M, N)
    if relu:
        ref = np.maximum(ref, 0)
    lib = _lib.lib()
    kind, flops = ctypes.c_int(-1), ctypes.c_double(0)
    _lib.check(lib.mpsr_conv2d_plan(M, 1, 1, K, N, 1, 1, 1, ctypes.byref(kind), ctypes.byref(flops)))
    assert kind.value == 6 and flops.value == 2.0 * M * K * N
    got = dn.conv2d(_dev(x), _dev(w), _dev(bias) if has_bias else None, _dev(res) if has_res else None, 1, 1, 1, relu,
                    split_k=1)
    _close(got.reshape(M, N), torch.from_numpy(ref), 1e-5, "few-row FC %s" % ((M, K, N),))
    again = dn.conv2d(_dev(x), _dev(w), _dev(bias) if has_bias else None, _dev(res) if has_res else None, 1, 1, 1, relu,
                      split_k=1)
    assert torch.equal(got, again), "not deterministic"
    lib.mpsr_debug_set_conv_pointwise(0)
    try:
        gemm = dn.conv2d(_dev(x), _dev(w), _dev(bias) if has_bias else None, _dev(res) if has_res else None, 1, 1, 1, relu,
                         split_k=1)
    finally:
        lib.mpsr_debug_set_conv_pointwise(-1)
    _close(got, gemm, 1e-5, "few-row FC vs implicit GEMM")


@pytest.mark.parametrize("B,dil,C,N,has_bias,relu,th", [
    (8, 4, 64, 64, True, True, 1), (3, 4, 32, 40, False, False, 1), (1, 4, 16, 4, True, False, 1),
    (5, 2, 48, 200, True, True, 1), (64, 4, 256, 256, True, True, 1), (7, 3, 16, 65, True, True, 1),
    (33, 4, 128, 128, True, True, 1),
    # tiles with halos: th x th tiles per pixel sub-grid (block2's conv2 is 12x12 at dilation 2 = 6x6 sub-grids)
    (64, 2, 128, 128, True, True, 2), (3, 2, 32, 40, False, False, 2), (2, 2, 16, 65, True, True, 3),
    (5, 3, 48, 24, True, False, 2), (1, 2, 16, 8, False, True, 4),
    # ... and dense 3x3 layers (dilation 1): block1's conv2 is 12x12 = 4 x 4 tiles
    (64, 1, 64, 64, True, True, 4), (3, 1, 32, 40, False, False, 2),
    # ... up to the decoder's maps at small batches: 24x24 = 8 x 8 tiles, 48x48 = 16 x 16
    (3, 1, 32, 40, True, True, 8), (2, 1, 64, 64, True, False, 16), (1, 2, 16, 24, True, True, 5)])
def test_conv2d_winograd3_atrous_vs_fp64(B, dil, C, N, has_bias, relu, th):
    """The Winograd F(3x3,3x3) kernel for atrous 3x3 layers whose pixel sub-grids are 3x3 (csrc/winograd3.hip: H = W =
    3 * dilation; block3's conv2 is 12x12 at dilation 4) or th x th tiles of 3x3 with halos (H = W = 3 * dilation * th;
    block2's conv2 is 12x12 at dilation 2): ragged tile / channel counts, the seven-position waves, the real layer shapes.  Against float64 (1e-5 of the tensor scale; measured ~5e-6), deterministic, and really another
    evaluation than the border-class implicit GEMM (which must agree to 1e-5 as well)."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    H = 3 * dil * th
    rng = np.random.default_rng(B * 1000 + C + N + dil + 7 * th)
    x = rng.standard_normal((B, H, H, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    ref = _conv_ref(x, w, bias, None, dil, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_winograd(3)
    try:
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
        again = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    _close(got, ref, 1e-5, "winograd F(3x3) atrous %s" % ((B, dil, C, N),))
    assert torch.equal(got, again), "not deterministic"
    lib.mpsr_debug_set_conv_winograd(0)
    try:
        direct = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    _close(got, direct, 1e-5, "winograd vs implicit GEMM")
    assert not torch.equal(got, direct) or C * N < 1024


@pytest.mark.parametrize("M,K,N,has_bias,relu", [(32, 18432, 2048, True, True), (5, 4608, 512, True, False),
                                                  (64, 18432, 2048, False, True), (33, 8192, 100, True, True),
                                                  (1, 18432, 2048, True, True)])
def test_fc_few_rows_long_k_split_vs_fp64(M, K, N, has_bias, relu):
    """img_fc at the reference's 32 boxes per image (csrc/pointwise.hip, fc_rows_kernel<1>): few rows, K = 18432 --
    the few-row kernel with K cut into slabs over blockIdx.y and a finishing launch that adds the partial tiles in order
    (+ bias, ReLU).  Against float64, against the stream-K kernel it replaces there (mpsr_debug_set_fc_split_rows(0)),
    ragged rows / columns, deterministic."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(M + K + N)
    x = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    ref = x.astype(np.float64) @ w.astype(np.float64).T + (bias.astype(np.float64) if has_bias else 0.0)
    if relu:
        ref = np.maximum(ref, 0)
    lib = _lib.lib()
    run = lambda: dn.conv2d(_dev(x).reshape(M, 1, 1, K), _dev(w), _dev(bias) if has_bias else None, None, 1, 1, 1, relu,
                            split_k=0).reshape(M, N)
    got = run()
    assert torch.equal(got, run())
    lib.mpsr_debug_set_fc_split_rows(0)
    try:
        other = run()
    finally:
        lib.mpsr_debug_set_fc_split_rows(96)
    scale = np.abs(ref).max()
    assert np.abs(got.cpu().numpy() - ref).max() <= 2e-6 * scale
    assert np.abs(other.cpu().numpy() - ref).max() <= 2e-6 * scale
    assert not torch.equal(got, other)  # (it really is the other kernel)


@pytest.mark.parametrize("B,H,Wd,C,N,dil,has_bias,relu", [
    (1, 40, 152, 256, 256, 4, True, True), (1, 40, 152, 128, 128, 2, False, True), (2, 16, 16, 64, 72, 1, True, False),
    (3, 8, 12, 128, 40, 2, True, True)])
def test_conv2d_winograd_f2x2_k_split_of_small_launches(B, H, Wd, C, N, dil, has_bias, relu):
    """csrc/winograd.hip, SPLIT instantiation of the eight-wave F(2x2,3x3) kernel: a launch that cannot fill the chip (the
    full-image trunk's atrous layers at ONE image: 96 workgroups) runs its channel range as 2 .. 8 slices whose partial
    outputs a finishing launch adds in slice order (+ bias, ReLU): every slice count against float64, deterministic."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B * 31 + C + N + dil)
    x = np.maximum(rng.standard_normal((B, H, Wd, C)), 0).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    ref = _conv_ref(x, w, bias, None, dil, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    outs = {}
    lib.mpsr_debug_set_conv_winograd(1)
    try:
        for slices in (0, 2, 4, -1):
            lib.mpsr_debug_set_wino3z_split(slices)
            run = lambda: dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
            outs[slices] = run()
            assert torch.equal(outs[slices], run())
            _close(outs[slices], ref, 1e-5, "F(2x2,3x3), %d K slices %s" % (slices, (B, H, Wd, C, N, dil)))
    finally:
        lib.mpsr_debug_set_wino3z_split(-1)
        lib.mpsr_debug_set_conv_winograd(-1)
    assert not torch.equal(outs[2], outs[0])  # (C >= 64: at least four channel steps, two slices exist)
    assert any(torch.equal(outs[-1], outs[k]) for k in (0, 2, 4))


@pytest.mark.parametrize("B,dil,C,N,has_bias,relu", [(32, 4, 256, 256, True, True), (8, 4, 64, 64, True, False),
                                                      (33, 4, 128, 132, False, True), (5, 2, 256, 40, True, True),
                                                      (1, 4, 256, 256, True, True)])
def test_conv2d_sixteen_product_k_split_of_small_launches(B, dil, C, N, has_bias, relu):
    """csrc/winograd3z.hip, SPLIT instantiation: a launch too small to fill the chip (the reference's 32 boxes per image)
    cuts its channel range into 2 .. 16 slices, each workgroup stores partial outputs, a finishing launch adds them in
    slice order + bias + ReLU.  Every slice count against float64 and against the unsplit kernel; deterministic; the
    automatic choice is one of them."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    H = 3 * dil
    rng = np.random.default_rng(B * 77 + C + N)
    x = np.maximum(rng.standard_normal((B, H, H, C)), 0).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    w_ok, _ = W.fold_conv(w)
    ref = _conv_ref(x, w, bias, None, dil, relu)
    lib = _lib.lib()
    outs = {}
    lib.mpsr_debug_set_conv_winograd(3)
    try:
        for slices in (0, 2, 4, 8, 16, -1):
            lib.mpsr_debug_set_wino3z_split(slices)
            run = lambda: dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
            outs[slices] = run()
            assert torch.equal(outs[slices], run()), "not deterministic at %d slices" % slices
            _close(outs[slices], ref, 2e-6, "sixteen products, %d K slices %s" % (slices, (B, dil, C, N)))
    finally:
        lib.mpsr_debug_set_wino3z_split(-1)
        lib.mpsr_debug_set_conv_winograd(-1)
    if C >= 64:  # (eight channel steps or more: two slices exist, and they add in another order than one K loop)
        assert not torch.equal(outs[2], outs[0])
    assert any(torch.equal(outs[-1], outs[k]) for k in (0, 2, 4, 8, 16))
    if (B, C) == (32, 256):
        assert torch.equal(outs[-1], outs[8])  # 32 workgroups -> 8 slices of four channel steps


@pytest.mark.parametrize("B,dil,C,N,has_bias,relu,masked", [
    (8, 4, 64, 64, True, True, False), (3, 4, 32, 40, False, False, False), (1, 4, 16, 4, True, False, False),
    (5, 2, 48, 200, True, True, False), (64, 4, 256, 256, True, True, False), (7, 3, 16, 65, True, True, False),
    (33, 4, 128, 128, True, True, False), (2, 1, 32, 129, True, False, False), (9, 4, 16, 300, False, True, False),
    # data-gradient launches of the training path: the ReLU mask of the layer applied in the store path
    (8, 4, 64, 64, False, False, True), (3, 4, 32, 40, False, False, True), (64, 4, 256, 256, False, False, True),
    (7, 3, 16, 65, False, False, True)])
def test_conv2d_winograd3_wave_owned_forms(B, dil, C, N, has_bias, relu, masked):
    """The three kernels of an atrous 3x3 layer whose pixel sub-grids are single tiles (mpsr_debug_set_wino3_form):
    0 = csrc/winograd3.hip (F(3x3,3x3), a tile's 25 positions shared by eight waves, exchange through LDS), 1 =
    csrc/winograd3w.hip (the same products with one wave owning all 25 positions of its 32-tile x 32-channel block: 400
    accumulators, lane-local output transform) -- same transformed operands, same accumulation order, same output
    transform: IDENTICAL BITS; 2 = csrc/winograd3z.hip, the default: the zero-padded tile in SIXTEEN products (rank 4 per
    dimension, wino3_transforms.h) -- another algorithm, held to 2e-6 of the tensor scale against float64 (measured
    3e-7: the error of a direct convolution; F(3x3,3x3) 3e-6).  Ragged tile and channel counts (partial 32-tile blocks,
    partial 128-channel workgroups), every dilation, with and without bias / ReLU / the training path's mask."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    H = 3 * dil
    rng = np.random.default_rng(B * 1000 + C + N + dil)
    x = rng.standard_normal((B, H, H, C)).astype(np.float32)
    w = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    act = np.maximum(rng.standard_normal((B, H, H, N)), 0).astype(np.float32)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    outs = []
    lib.mpsr_debug_set_conv_winograd(3)
    try:
        for form in (0, 1, 2, -1):
            lib.mpsr_debug_set_wino3_form(form)
            if masked:
                ws = torch.empty((lib.mpsr_conv2d_scratch_floats(B, H, H, N),), dtype=torch.float32, device="cuda")
                got = torch.full((B, H, H, N), 7.0, dtype=torch.float32, device="cuda")
                xd, wd, ad = _dev(x), _dev(w_ok), _dev(act)
                _lib.check(lib.mpsr_conv2d_relu_masked_f32(_lib.ptr(xd), B, H, H, C, _lib.ptr(wd), _lib.ptr(ad),
                                                           _lib.ptr(got), N, 3, 3, dil, _lib.ptr(ws), ws.numel(),
                                                           _lib.stream()))
            else:
                got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, None, 3, 3, dil, relu, split_k=0)
            outs.append(got)
    finally:
        lib.mpsr_debug_set_wino3_form(-1)
        lib.mpsr_debug_set_conv_winograd(-1)
    assert torch.equal(outs[0], outs[1]), "the two F(3x3,3x3) kernels differ"
    assert torch.equal(outs[2], outs[3]), "the default is not the sixteen-product form"
    assert not torch.equal(outs[2], outs[0]) or C * N < 1024  # it really is another evaluation
    ref = _conv_ref(x, w, bias, None, dil, relu)
    if masked:
        ref = torch.where(torch.from_numpy(act) > 0, ref, torch.zeros_like(ref))
        assert 0.3 < float((outs[1] != 0).float().mean()) < 0.7
    _close(outs[1], ref, 1e-5, "winograd F(3x3,3x3) wave-owned %s" % ((B, dil, C, N),))
    _close(outs[2], ref, 2e-6, "sixteen-product form %s" % ((B, dil, C, N),))
    # the plan counts the products of the kernel the call takes
    kind, ex = ctypes.c_int(-1), ctypes.c_double(0)
    lib.mpsr_debug_set_conv_winograd(3)
    try:
        _lib.check(lib.mpsr_conv2d_plan(B, H, H, C, N, 3, 3, dil, ctypes.byref(kind), ctypes.byref(ex)))
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    assert kind.value == 4 and ex.value == 2.0 * B * dil * dil * 16 * C * N


def test_border_class_tiling_is_bit_identical():
    """Skipping the all-zero taps of an atrous layer must not change a single bit (same products, same order)."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(99)
    for (B, H, C, N, d) in ((5, 12, 64, 96, 4), (3, 12, 32, 64, 2), (2, 9, 16, 32, 4), (2, 8, 16, 32, 4),
                            (2, 7, 16, 32, 4), (3, 24, 16, 32, 1)):
        x = _dev(rng.standard_normal((B, H, H, C)).astype(np.float32))
        w = _dev((rng.standard_normal((N, 9 * C)) / np.sqrt(9 * C)).astype(np.float32))
        outs = []
        for mode in (0, 1):
            _lib.lib().mpsr_debug_set_conv_classes(mode)
            try:
                outs.append(dn.conv2d(x, w, None, None, 3, 3, d, False).cpu().numpy())
            finally:
                _lib.lib().mpsr_debug_set_conv_classes(-1)
        np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("tile,depth", [(-1, -1), (3, 1), (3, 2), (0, 1), (5, 2)])
def test_class_tap_instantiation_is_bit_identical(tile, depth):
    """3x3 layers tiled by border classes with C % 32 == 0 run an instantiation whose loads carry the tap offset in a
    scalar register and no per-load bounds test (every tap of a class is in-image for all its rows): same products,
    same order as the general kernel."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(321)
    lib = _lib.lib()
    for (B, H, Wd, C, N, d) in ((5, 12, 12, 64, 96, 4), (3, 12, 12, 32, 64, 2), (2, 9, 11, 32, 36, 4),
                                (2, 8, 8, 64, 32, 4), (3, 24, 10, 32, 32, 1), (1, 2, 2, 32, 8, 1),
                                (2, 7, 7, 32, 32, 4)):  # the last: H < 2 * dilation, no classes on that axis
        x = _dev(rng.standard_normal((B, H, Wd, C)).astype(np.float32))
        w = _dev((rng.standard_normal((N, 9 * C)) / np.sqrt(9 * C)).astype(np.float32))
        bias = _dev(rng.standard_normal(N).astype(np.float32))
        outs = []
        lib.mpsr_debug_set_conv_tile(tile)
        lib.mpsr_debug_set_conv_depth(depth)
        lib.mpsr_debug_set_conv_classes(1)
        try:
            for mode in (0, -1):
                lib.mpsr_debug_set_conv_plain(mode)
                outs.append(dn.conv2d(x, w, bias, None, 3, 3, d, True).cpu().numpy())
        finally:
            lib.mpsr_debug_set_conv_plain(-1)
            lib.mpsr_debug_set_conv_classes(-1)
            lib.mpsr_debug_set_conv_tile(-1)
            lib.mpsr_debug_set_conv_depth(-1)
        np.testing.assert_array_equal(outs[0], outs[1])
        ref = _conv_ref(x.cpu().numpy(), w.cpu().numpy().reshape(N, 3, 3, C).transpose(1, 2, 3, 0), bias.cpu().numpy(),
                        None, d, True)
        np.testing.assert_allclose(outs[1], ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("tile,depth", [(-1, -1), (3, 1), (3, 2), (5, 1), (5, 2), (0, 1)])
def test_plain_1x1_instantiation_is_bit_identical(tile, depth):
    """1x1 layers with C % 32 == 0 run an instantiation without row decode / tap logic (scalar K-step offsets);
    same products in the same order as the general kernel: not a bit may differ, ragged M and N tails, bias,
    residual, ReLU and split-K included."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(123)
    lib = _lib.lib()
    for (B, H, Wd, C, N, split) in ((3, 12, 12, 256, 96, 1), (2, 7, 5, 64, 130, 1), (37, 1, 1, 512, 24, 1),
                                    (2, 12, 12, 128, 64, 2), (1, 3, 3, 32, 4, 1)):
        x = _dev(rng.standard_normal((B, H, Wd, C)).astype(np.float32))
        w = _dev((rng.standard_normal((N, C)) / np.sqrt(C)).astype(np.float32))
        bias = _dev(rng.standard_normal(N).astype(np.float32))
        res = _dev(rng.standard_normal((B, H, Wd, N)).astype(np.float32))
        outs = []
        lib.mpsr_debug_set_conv_tile(tile)
        lib.mpsr_debug_set_conv_depth(depth)
        try:
            for mode in (0, -1):
                lib.mpsr_debug_set_conv_plain(mode)
                outs.append(dn.conv2d(x, w, bias, res, 1, 1, 1, True, split_k=split).cpu().numpy())
        finally:
            lib.mpsr_debug_set_conv_plain(-1)
            lib.mpsr_debug_set_conv_tile(-1)
            lib.mpsr_debug_set_conv_depth(-1)
        np.testing.assert_array_equal(outs[0], outs[1])
        ref = np.maximum(x.cpu().numpy().reshape(-1, C).astype(np.float64) @ w.cpu().numpy().T.astype(np.float64)
                         + bias.cpu().numpy() + res.cpu().numpy().reshape(-1, N), 0)
        np.testing.assert_allclose(outs[1].reshape(-1, N), ref, rtol=0, atol=2e-5 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("split", [2, 3, 8, 16])
def test_fc_split_k(split):
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(split)
    B, K, N = 37, 1152, 200
    x = rng.standard_normal((B, 1, 1, K)).astype(np.float32)
    w = (rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    ref = np.maximum(x.reshape(B, K).astype(np.float64) @ w.T.astype(np.float64) + bias, 0)
    got = dn.conv2d(_dev(x), _dev(w), _dev(bias), None, 1, 1, 1, True, split_k=split)
    _close(got.reshape(B, N), ref, 2e-6, "split_k")


def test_conv2d_argument_errors():
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    x = torch.zeros((1, 4, 4, 6), device="cuda")
    with pytest.raises(_lib.InvalidArgumentError):
        dn.conv2d(x, torch.zeros((8, 6), device="cuda"))           # C % 4 != 0
    x = torch.zeros((1, 4, 4, 8), device="cuda")
    with pytest.raises(_lib.InvalidArgumentError):
        dn.conv2d(x, torch.zeros((8, 32), device="cuda"), kh=2, kw=2)  # even kernel


# ------------------------------------------------------------------------------------------- network pieces

def _inputs(B, seed=0):
    rng = np.random.default_rng(seed)
    crops = (rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)
    y1 = rng.uniform(0, 150, B)
    x1 = rng.uniform(0, 1000, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(20, 200, B), x1 + rng.uniform(20, 200, B)], 1).astype(np.float32)
    cam_p = np.array([[721.5377, 0.0, 609.5593, 44.85728], [0.0, 721.5377, 172.854, 0.2163791],
                      [0.0, 0.0, 1.0, 0.002745884]], np.float32)  # a KITTI P2
    view = rng.uniform(-0.6, 0.6, B).astype(np.float32)
    cls = np.ones((B, 1), np.int32)
    mean_lwh = np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))
    z_off = np.full((B,), 2.17799973487854, np.float32)
    return crops, boxes, cam_p, view, cls, mean_lwh, z_off


@pytest.mark.parametrize("width_div,B", [(2, 3), (1, 2)])
def test_trunk_vs_oracle(width_div, B):
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    weights = W.synthetic_weights(seed=3, width_div=width_div, decoder=False, heads=False)
    crops = _inputs(B)[0]
    ref = onet.resnet101_block3(torch.from_numpy(crops), weights, W.CROP_SCOPE)
    net = dn.DeviceNet.__new__(dn.DeviceNet)
    net.device = torch.device("cuda")
    net.crop_trunk = dn.PackedPart(*W.pack_trunk(weights, W.CROP_SCOPE, width_div), net.device)
    net.full_trunk = None
    net.ws_trunk = dn.Workspace(net.device)
    got = net.trunk(_dev(crops))
    assert tuple(got.shape) == (B, 12, 12, 1024 // width_div)
    _close(got, ref, 1e-4, "trunk block3")


def test_trunk_full_image_shape():
    """The same entry point serves the full-image branch (1,160,608,3) -> (1,40,152,C)."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    weights = W.synthetic_weights(seed=4, width_div=2, decoder=False, heads=False)
    img = (np.random.default_rng(4).standard_normal((1, 160, 608, 3)) * 50).astype(np.float32)
    ref = onet.resnet101_block3(torch.from_numpy(img), weights, W.CROP_SCOPE)
    net = dn.DeviceNet.__new__(dn.DeviceNet)
    net.device = torch.device("cuda")
    net.crop_trunk = dn.PackedPart(*W.pack_trunk(weights, W.CROP_SCOPE, 2), net.device)
    net.full_trunk = None
    net.ws_trunk = dn.Workspace(net.device)
    got = net.trunk(_dev(img))
    assert tuple(got.shape) == (1, 40, 152, 512)
    _close(got, ref, 1e-4, "full-image trunk")


@pytest.mark.parametrize("width_div,B", [(2, 3), (1, 2)])
def test_squash_decoder_and_heads_vs_oracle(width_div, B):
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    weights = W.synthetic_weights(seed=5, width_div=width_div, trunk=False)
    rng = np.random.default_rng(6)
    cf = 1024 // width_div
    crop_feat = np.maximum(rng.standard_normal((B, 12, 12, cf)), 0).astype(np.float32)
    full_feat = np.maximum(rng.standard_normal((B, 12, 12, cf)), 0).astype(np.float32)
    _, boxes, cam_p, view, cls, mean_lwh, z_off = _inputs(B, 7)
    rb, rm, rx = onet.squash_decoder(torch.from_numpy(crop_feat), torch.from_numpy(full_feat), weights, 48, 48)
    rh = onet.heads(rb, boxes, cam_p, view, cls, mean_lwh, z_off, weights)
    net = dn.DeviceNet.__new__(dn.DeviceNet)
    net.device = torch.device("cuda")
    net.decoder = dn.PackedPart(*W.pack_decoder(weights, width_div), net.device)
    feat_elems = weights["output/proposal_fc/proposal_fc/img_fc/weights"].shape[0]
    net.heads = dn.PackedPart(*W.pack_heads(weights, feat_elems), net.device)
    net.ws_dec, net.ws_heads = dn.Workspace(net.device), dn.Workspace(net.device)
    gb, gm, gx = net.squash_decoder(_dev(crop_feat), _dev(full_feat), (48, 48))
    _close(gb, rb, 1e-4, "features_for_box_3d")
    _close(gm, rm, 1e-4, "features_for_map")
    _close(gx, rx, 1e-4, "inst_xyz_map_local")
    gh = net.heads_fwd(gb, _dev(boxes), _dev(cam_p), _dev(view), _dev(cls), _dev(mean_lwh), _dev(z_off))
    for key in ("lwh", "lwh_offs", "alpha_bins", "alpha_regs", "prop_cen_z", "cen_y", "cen_y_offs", "cen_z",
                "cen_z_offs", "cen_x", "centroids", "view_ang"):
        _close(gh[key], rh[key], 1e-4, key)


def test_instance_path_end_to_end():
    """Proposal crops in -> centroids + N x 3 local cloud out, full width, B = 2, vs the CPU restatement; then
    Chamfer of the predicted cloud against a synthetic GT cloud on both paths (idx bit-exact when the predicted
    clouds agree to the last bit is not required: the clouds differ by float drift, so Chamfer is compared to 1e-4)."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance
    from oracle import ops as orc
    B = 2
    weights = W.synthetic_weights(seed=11)
    crops, boxes, cam_p, view, cls, mean_lwh, z_off = _inputs(B, 12)
    rng = np.random.default_rng(13)
    full_feat = np.maximum(rng.standard_normal((B, 12, 12, 1024)), 0).astype(np.float32)
    ref = onet.instance_path(crops, full_feat, boxes, cam_p, view, cls, mean_lwh, z_off, weights)
    net = dn.DeviceNet(weights)
    crop_feat = net.trunk(_dev(crops))
    fb, fm, xyz = net.squash_decoder(crop_feat, _dev(full_feat), (48, 48))
    out = net.heads_fwd(fb, _dev(boxes), _dev(cam_p), _dev(view), _dev(cls), _dev(mean_lwh), _dev(z_off))
    _close(crop_feat, ref["crop_feat"], 1e-4, "crop_feat")
    _close(xyz, ref["inst_xyz_map_local"], 1e-4, "inst_xyz_map_local")
    _close(out["centroids"], ref["centroids"], 1e-4, "centroids")
    _close(out["lwh"], ref["lwh"], 1e-4, "lwh")
    gt = rng.standard_normal((B, 2304, 3)).astype(np.float32)
    d1, _, d2, _ = tf_nndistance.nn_distance(xyz.reshape(B, -1, 3), _dev(gt))
    r1, _, r2, _ = orc.nn_distance(ref["inst_xyz_map_local"].reshape(B, -1, 3).numpy(), gt)
    np.testing.assert_allclose((d1.sum(1) + d2.sum(1)).cpu().numpy(), r1.sum(1) + r2.sum(1), rtol=1e-4)


# ------------------------------------------------------------------------------------------- reference-API mirrors

def _sample(B, seed):
    crops, boxes, cam_p, view, cls, mean_lwh, z_off = _inputs(B, seed)
    return dict(boxes_2d=_dev(boxes), cam_p=_dev(cam_p), est_view_angs=_dev(view), class_indices=_dev(cls),
                mean_lwh=_dev(mean_lwh), prop_cen_z_offset=_dev(z_off)), (crops, boxes, cam_p, view, cls, mean_lwh,
                                                                           z_off)


def test_model_build_fused_vs_output_builder_vs_oracle():
    """MonoPSRModel.build on pre-cropped inputs: the fused native heads and the method-by-method
    MonoPSROutputBuilder mirror give the same output_dict, and both match the CPU restatement."""
    from monopsr_amd.core import config_utils, constants
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel
    B = 3
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=21, width_div=2)
    net = dn.DeviceNet(weights, width_div=2)
    sample, (crops, boxes, cam_p, view, cls, mean_lwh, z_off) = _sample(B, 22)
    full_feat = np.maximum(np.random.default_rng(23).standard_normal((B, 12, 12, 512)), 0).astype(np.float32)
    sample['rgb_image_crops'] = _dev(crops)
    sample['full_img_feature_crop'] = _dev(full_feat)
    ref = onet.instance_path(crops, full_feat, boxes, cam_p, view, cls, mean_lwh, z_off, weights)
    outs = []
    for fused in (True, False):
        model = MonoPSRModel(cfg.model_config, cfg.dataset_config, net, 'test', fused_heads=fused)
        out, feats = model.build(dict(sample))
        outs.append(out)
        assert tuple(feats[constants.FEATURES_FOR_MAP].shape) == (B, 48, 48, 64)
        assert tuple(feats[constants.FEATURES_FOR_BOX_3D].shape) == (B, 6, 6, 256)
        for key in ("inst_xyz_map_local", "lwh", "lwh_offs", "alpha_bins", "alpha_regs", "view_ang", "prop_cen_z",
                    "cen_y", "cen_y_offs", "cen_z", "cen_z_offs", "cen_x", "centroids"):
            _close(out[key], ref[key], 1e-4, "%s (fused=%s)" % (key, fused))
    _close(outs[0]["centroids"], outs[1]["centroids"], 1e-5, "fused vs builder")


def test_model_build_full_image_path_vs_oracle():
    """image in -> preprocess, proposal crops, both trunks, feature crop + pool -> outputs."""
    from monopsr_amd.core import config_utils
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel
    B = 3
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=31, width_div=2, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    net = dn.DeviceNet(weights, width_div=2, full_trunk=True)
    rng = np.random.default_rng(32)
    H, Wd = 375, 1242
    rgb = rng.integers(0, 256, (H, Wd, 3)).astype(np.float32)
    sample, (_, boxes, cam_p, view, cls, mean_lwh, z_off) = _sample(B, 33)
    boxes[:, 2] = np.minimum(boxes[:, 2], H - 1)
    boxes[:, 3] = np.minimum(boxes[:, 3], Wd - 1)
    norm = (boxes / np.array([H, Wd, H, Wd], np.float32)).astype(np.float32)  # kitti_dataset.py:450
    sample['boxes_2d'] = _dev(boxes)
    sample['rgb_image'] = _dev(rgb)
    sample['boxes_2d_norm'] = _dev(norm)
    ref = onet.full_image_path(rgb, boxes, norm, cam_p, view, cls, mean_lwh, z_off, weights)
    model = MonoPSRModel(cfg.model_config, cfg.dataset_config, net, 'test')
    out, _ = model.build(sample)
    for key in ("inst_xyz_map_local", "lwh", "alpha_bins", "centroids"):
        _close(out[key], ref[key], 1e-4, key)


def test_model_build_batch_of_images_equals_single_image_calls_and_oracle(golden_dir):
    """MonoPSRModel.build_batch / DeviceNet.forward_images: the reference's step (one image + its boxes,
    monopsr_model.py:222-237, net_builder.py:44-60, configs/monopsr_model_000.yaml:14-17) for N = 8 images in one pass --
    one full-image trunk call with batch 8, box_ind-routed proposal and feature crops, one crop-trunk / decoder / heads
    call over all boxes, a projection matrix per image.  Frames: the real KITTI frame of the cfg1 fixture (with its own
    label boxes), variants of it at other sizes, and synthetic frames; box counts differ per image.  Every image's
    outputs must equal its single-image `build` (<= 2e-5 of the tensor scale) and the CPU restatement (<= 1e-4)."""
    import os
    from monopsr_amd.core import config_utils
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel
    fx = np.load(os.path.join(golden_dir, "kitti_cfg1.npz"))
    frame = fx["full_frame"].astype(np.float32)
    rows = np.nonzero(fx["frame_index"] == int(fx["full_frame_index"]))[0]
    cfg = config_utils.default_config()
    weights = W.synthetic_weights(seed=31, width_div=2, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    net = dn.DeviceNet(weights, width_div=2, full_trunk=True)
    model = MonoPSRModel(cfg.model_config, cfg.dataset_config, net, 'test')
    rng = np.random.default_rng(77)
    frames = [frame, frame[3:, 8:].copy(), frame[:, ::-1].copy(), frame[:-5, :-18].copy()]
    frames += [rng.integers(0, 256, (375, 1242, 3)).astype(np.float32) for _ in range(4)]
    counts = [len(rows), 5, 2, 7, 3, 1, 6, 4]
    samples, host = [], []
    for i, (img, nb) in enumerate(zip(frames, counts)):
        H, Wd = img.shape[:2]
        sample, (_, boxes, cam_p, view, cls, mean_lwh, z_off) = _sample(nb, 100 + i)
        if i == 0:
            boxes = fx["boxes_2d"][rows].astype(np.float32)
            cam_p = fx["cam_p"][rows[0]].astype(np.float32)
        else:
            cam_p = (cam_p + rng.uniform(-2, 2, cam_p.shape) * (np.abs(cam_p) > 1)).astype(np.float32)
        boxes[:, 2] = np.minimum(boxes[:, 2], H - 1)
        boxes[:, 3] = np.minimum(boxes[:, 3], Wd - 1)
        norm = (boxes / np.array([H, Wd, H, Wd], np.float32)).astype(np.float32)  # kitti_dataset.py:450
        sample.update(boxes_2d=_dev(boxes), boxes_2d_norm=_dev(norm), rgb_image=_dev(img), cam_p=_dev(cam_p))
        samples.append(sample)
        host.append((img, boxes, norm, cam_p, view, cls, mean_lwh, z_off))
    batch = model.build_batch([dict(s) for s in samples])
    assert len(batch) == len(samples)
    keys = ("inst_xyz_map_local", "lwh", "lwh_offs", "alpha_bins", "alpha_regs", "prop_cen_z", "cen_y", "cen_z", "cen_x",
            "centroids", "view_ang")
    for i, s in enumerate(samples):
        single, _ = model.build(dict(s))
        for key in keys:
            assert batch[i][key].shape == single[key].shape, (i, key)
            _close(batch[i][key], single[key], 2e-5, "image %d %s (batch vs single)" % (i, key))
    for i in (0, 3):  # the real frame and a resized-input variant against the oracle
        ref = onet.full_image_path(*host[i], weights)
        for key in ("inst_xyz_map_local", "lwh", "alpha_bins", "centroids"):
            _close(batch[i][key], ref[key], 1e-4, "image %d %s (vs oracle)" % (i, key))
    assert model.build_batch([]) == []
    # an image WITHOUT boxes between images with boxes (reshape(0, -1) used to raise), and only such images
    def no_boxes(s):
        e = dict(s)
        for k in ("boxes_2d", "boxes_2d_norm", "est_view_angs", "class_indices", "mean_lwh", "prop_cen_z_offset"):
            e[k] = s[k][:0]
        return e
    mixed = model.build_batch([dict(samples[1]), no_boxes(samples[2]), dict(samples[3])])
    assert mixed[1]["centroids"].shape == (0, 3) and mixed[1]["inst_xyz_map_local"].shape[0] == 0
    for j, i in ((0, 1), (2, 3)):
        for key in keys:
            _close(mixed[j][key], batch[i][key], 2e-5, "image %d %s (with an empty image in the call)" % (i, key))
    none = model.build_batch([no_boxes(samples[0]), no_boxes(samples[1])])
    assert [o["centroids"].shape for o in none] == [(0, 3), (0, 3)] and none[0]["inst_xyz_map_local"].shape == (0, 48, 48, 3)


def test_heads_with_a_projection_matrix_per_box_equal_per_image_calls():
    """mpsr_heads_fwd_cams: cam_index routes every box to its image's projection matrix; the concatenated call gives each
    box the bits of its own single-matrix call (same FC kernels row by row when the batch sizes pick the same kernel;
    here both are the few-row FC path)."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    weights = W.synthetic_weights(seed=5, width_div=4)
    net = dn.DeviceNet(weights, width_div=4)
    rng = np.random.default_rng(6)
    nimg, per = 3, 4
    B = nimg * per
    feat = _dev(np.maximum(rng.standard_normal((B, 6, 6, 128)), 0).astype(np.float32))
    _, boxes, cam_p, view, cls, mean_lwh, z_off = _inputs(B, 7)
    cams = np.stack([cam_p * (1 + 0.01 * k) for k in range(nimg)]).astype(np.float32)
    idx = np.repeat(np.arange(nimg), per).astype(np.int32)
    got = net.heads_fwd(feat, _dev(boxes), _dev(cams), _dev(view), _dev(cls), _dev(mean_lwh), _dev(z_off),
                        cam_index=_dev(idx))
    for k in range(nimg):
        sl = slice(k * per, (k + 1) * per)
        one = net.heads_fwd(feat[sl], _dev(boxes[sl]), _dev(cams[k]), _dev(view[sl]), _dev(cls[sl]), _dev(mean_lwh[sl]),
                            _dev(z_off[sl]))
        for key in ("centroids", "lwh", "alpha_bins", "alpha_regs", "cen_x", "prop_cen_z"):
            _close(got[key][sl], one[key], 1e-5, "%s image %d" % (key, k))
    with pytest.raises(ValueError):
        net.heads_fwd(feat, _dev(boxes), _dev(cams), _dev(view), _dev(cls), _dev(mean_lwh), _dev(z_off))
    # hostile cam_index: negative / past n_cams is never dereferenced -- those boxes' centroids are NaN, the
    # other boxes keep their bits, and outputs that do not depend on the projection matrix stay finite
    bad = idx.copy()
    bad[1], bad[6] = -1, nimg + 1000000
    hostile = net.heads_fwd(feat, _dev(boxes), _dev(cams), _dev(view), _dev(cls), _dev(mean_lwh), _dev(z_off),
                            cam_index=_dev(bad))
    ok = np.ones(B, bool)
    ok[[1, 6]] = False
    cen = hostile["centroids"].cpu().numpy()
    assert np.isnan(cen[~ok]).any(axis=1).all() and np.isfinite(cen[ok]).all()
    assert torch.equal(hostile["centroids"][torch.from_numpy(ok).cuda()], got["centroids"][torch.from_numpy(ok).cuda()])


def test_evaluate_predictions_metrics():
    """Chamfer / EMD metric wiring (monopsr_model.py:1112-1170) vs the oracle ops on the same masked clouds."""
    from monopsr_amd.core import config_utils, constants
    from monopsr_amd.core.models.monopsr.monopsr_model import MonoPSRModel
    from oracle import ops as orc
    cfg = config_utils.default_config()
    model = MonoPSRModel(cfg.model_config, cfg.dataset_config, None, 'val')
    rng = np.random.default_rng(41)
    B = 4
    pred = rng.uniform(-1, 1, (B, 16, 16, 3)).astype(np.float32)
    gt = rng.uniform(-1, 1, (B, 16, 16, 3)).astype(np.float32)
    mask = (rng.uniform(size=(B, 16, 16, 1)) > 0.25).astype(np.float32)
    m = model.evaluate_predictions({constants.KEY_INST_XYZ_MAP_LOCAL: _dev(pred)},
                                   {constants.KEY_INST_XYZ_MAP_LOCAL: _dev(gt),
                                    constants.KEY_VALID_MASK_MAPS: _dev(mask)}, num_objs=3)
    p, g = (pred * mask).reshape(B, -1, 3), (gt * mask).reshape(B, -1, 3)
    nvalid = mask.reshape(B, -1).sum(1)
    d1, _, d2, _ = orc.nn_distance(p, g)
    np.testing.assert_allclose(m[constants.METRIC_CHAMFER].cpu().numpy(), ((d1.sum(1) + d2.sum(1)) / nvalid)[:3],
                               rtol=1e-5)
    emd = orc.match_cost(p, g, orc.approx_match(p, g, "gpu"), "gpu")
    np.testing.assert_allclose(m[constants.METRIC_EMD].cpu().numpy(), (emd / nvalid)[:3], rtol=1e-3)


def test_device_net_batch_chunking_is_bit_identical():
    """Batches above DeviceNet.MAX_CHUNK run as several native calls; instances are independent, and as long as the
    library picks the SAME kernel for every chunk each output element's K order is unchanged: the same bits.  (Kernel
    choice depends on the chunk's size -- the pointwise kernel wants >= 512 tiles, F(3x3,3x3) >= 1024 sub-grids -- so a
    short last chunk of a real batch can differ from its neighbours in the last bits; here every chunk is small.)"""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    weights = W.synthetic_weights(seed=71, width_div=4)
    net = dn.DeviceNet(weights, width_div=4)
    rng = np.random.default_rng(72)
    B = 5
    crops = _dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32))
    full = _dev(np.maximum(rng.standard_normal((B, 12, 12, 256)), 0).astype(np.float32))
    f1 = net.trunk(crops)
    a1 = net.squash_decoder(f1, full)
    old = dn.DeviceNet.MAX_CHUNK
    dn.DeviceNet.MAX_CHUNK = 2
    try:
        f2 = net.trunk(crops)
        a2 = net.squash_decoder(f2, full)
    finally:
        dn.DeviceNet.MAX_CHUNK = old
    assert torch.equal(f1, f2)
    for x, y in zip(a1, a2):
        assert torch.equal(x, y)


# ------------------------------------------------------------------------------- opt-in bf16x3 contraction mode

@pytest.fixture
def bf16x3():
    from monopsr_amd import _lib
    prev = _lib.set_conv_math("bf16x3")
    yield
    _lib.set_conv_math(prev)
    _lib.lib().mpsr_debug_set_conv_tile(-1)
    _lib.lib().mpsr_debug_set_conv_classes(-1)


@pytest.mark.parametrize("tile", [-1, 0, 1, 2, 3, 4, 5])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_bf16x3_vs_fp64(case, tile, bf16x3):
    """Split-bfloat16 products (hi*hi + hi*lo + lo*hi, fp32 accumulate): every tile instantiation within 4e-5 of
    the float64 convolution (the fp32 path: 2e-6)."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    B, H, Wd, C, N, k, rate, has_bias, has_res, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((k, k, C, N)) / np.sqrt(k * k * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
    ref = _conv_ref(x, w, bias, res, rate, relu)
    w_ok, _ = W.fold_conv(w)
    _lib.lib().mpsr_debug_set_conv_tile(tile)
    _lib.lib().mpsr_debug_set_conv_classes(1 if rate > 1 else -1)
    got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, _dev(res) if has_res else None, k, k, rate,
                    relu)
    _close(got, ref, 4e-5, "bf16x3 conv %s tile %d" % (case, tile))


def test_network_drift_in_bf16x3_mode(bf16x3):
    """Whole instance path (full-width trunk, decoder, heads) in bf16x3 mode vs the fp32 CPU restatement: within
    2e-4 of tensor scale (measured 1e-5..4e-5), a fifth of the path's 1e-3 budget; fp32 mode: 1e-5."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    B = 4
    weights = W.synthetic_weights(seed=0)
    sample, (crops, boxes, cam_p, view, cls, mean_lwh, z_off) = _sample(B, 5)
    full_feat = np.maximum(np.random.default_rng(6).standard_normal((B, 12, 12, 1024)), 0).astype(np.float32)
    ref = onet.instance_path(crops, full_feat, boxes, cam_p, view, cls, mean_lwh, z_off, weights)
    net = dn.DeviceNet(weights)
    feat = net.trunk(_dev(crops))
    fb, fm, xyz = net.squash_decoder(feat, _dev(full_feat))
    out = net.heads_fwd(fb, _dev(boxes), _dev(cam_p), _dev(view), _dev(cls), _dev(mean_lwh), _dev(z_off))
    worst = 0.0
    for got, key in ((feat, "crop_feat"), (xyz, "inst_xyz_map_local"), (out["centroids"], "centroids"),
                     (out["lwh"], "lwh"), (out["alpha_bins"], "alpha_bins")):
        want = ref[key].numpy()
        err = np.abs(got.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-30)
        assert err < 2e-4, "%s drift %.3e" % (key, err)
        worst = max(worst, err)
    assert worst > 1e-7   # the mode really was active (fp32 mode sits at ~3e-6 on block3 but ~1e-7 on lwh)


@pytest.mark.parametrize("split", [0, 2, 3, 5])
@pytest.mark.parametrize("case", [
    (2, 12, 12, 256, 256, 3, 4, True, False, True),    # block3 conv2: border classes + split-K together
    (1, 12, 12, 128, 128, 3, 2, True, False, True),
    (2, 12, 12, 1024, 256, 1, 1, True, False, True),   # block3 conv1 at a small batch
    (2, 12, 12, 256, 1024, 1, 1, True, True, True),    # conv3 + residual
    (1, 9, 11, 36, 40, 3, 2, False, True, False),      # ragged everything
    (1, 5, 5, 64, 64, 3, 4, True, False, False),       # dilation close to the map size: single-class fallback
])
def test_conv2d_split_k_with_classes_and_auto(case, split):
    """split_k > 1 on convolutions (slices per pixel class, empty slices write zero partials) and split_k = 0 (auto:
    the library fills an under-filled launch) give the same result as the unsplit kernel up to summation order."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    B, H, Wd, C, N, k, rate, has_bias, has_res, relu = case
    rng = np.random.default_rng(hash(case) % (2 ** 31))
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((k, k, C, N)) / np.sqrt(k * k * C)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
    res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
    ref = _conv_ref(x, w, bias, res, rate, relu)
    w_ok, _ = W.fold_conv(w)
    for classes in (-1, 0, 1):
        _lib.lib().mpsr_debug_set_conv_classes(classes)
        try:
            got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias) if has_bias else None, _dev(res) if has_res else None, k, k,
                            rate, relu, split_k=split)
        finally:
            _lib.lib().mpsr_debug_set_conv_classes(-1)
        _close(got, ref, 2e-6, "split %d classes %d %s" % (split, classes, case))

@pytest.mark.parametrize("B", [43, 171])
def test_network_with_and_without_the_pointwise_kernels(B):
    """The persistent pointwise / few-row FC kernels (csrc/pointwise.hip) take different subsets of the 1x1 layers at
    different batch sizes (>= 512 tiles per layer: conv3 from 43 crops on, conv1 from 171); the network's outputs must
    agree with the implicit-GEMM-only run to summation-order accuracy at any of them."""
    import bench
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    dev = torch.device("cuda", 0)
    net = dn.DeviceNet(W.synthetic_weights(seed=0), device=dev)
    inp, _ = bench.make_inputs(B, 256, 0, dev)
    step = bench.Step(net, inp, 256)
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_pointwise(0)
    try:
        xyz0, out0 = step.forward_net()
        xyz0, cen0 = xyz0.clone(), out0["centroids"].clone()
    finally:
        lib.mpsr_debug_set_conv_pointwise(-1)
    xyz1, out1 = step.forward_net()
    _close(xyz1, xyz0, 5e-5, "xyz map, B = %d" % B)
    _close(out1["centroids"], cen0, 5e-5, "centroids, B = %d" % B)
    assert not torch.equal(xyz1, xyz0)  # it really took other kernels


def test_scheduler_scratch_is_bounded_over_streams():
    """The per-(device, stream) scratch of library-scheduled launches is kept for at most eight streams
    (device_net.stream_scratch): short-lived streams must not pin ~100 MB each for the life of the process."""
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(3)
    x = _dev(rng.standard_normal((2, 12, 12, 64)).astype(np.float32))
    w = _dev((rng.standard_normal((64, 64)) / 8).astype(np.float32))
    ref = dn.conv2d(x, w, None, None, 1, 1, 1, True, split_k=0)
    for _ in range(12):
        st = torch.cuda.Stream()
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            got = dn.conv2d(x, w, None, None, 1, 1, 1, True, split_k=0)
        st.synchronize()
        assert torch.equal(got, ref)
    assert len(dn._SCHED_SCRATCH) <= 8
