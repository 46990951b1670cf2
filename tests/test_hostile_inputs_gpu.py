"""Parity of the transform-domain kernels on HOSTILE inputs (r03 review, weak 1a / item 6): every other float tolerance
in this suite is max|err| / max|ref| on N(0,1)-like tensors, which says nothing about a heavy-tailed activation map or
about the small outputs next to a large one.  Here the operands look like a trained network's worst case -- post-ReLU
maps with 90 % zeros and 1 % of the entries scaled by 1e3, filters whose per-output-channel scale spreads over two
decades (BatchNorm-folded weights) -- and the bound is ELEMENT-WISE as well: the relative error of every element with
|ref| > 1e-3 max|ref|, next to the usual tensor-scale bound.  References: float64 restatements (oracle/net.py
operators), never the library itself.

Measured (MI355X, r04) and asserted: the direct kernels, the pointwise kernel and the upsampled-conv tap GEMM keep every
such element within 1e-3 of ITSELF (measured 1e-5 .. 2e-4).  The Winograd kernels do not on 1000x outliers: their
transforms mix a whole patch, so a small output next to a huge activation carries an error relative to the huge one --
F(4x4,3x3) 2e-3 .. 4e-3, F(3x3,3x3) 2e-3, bounded here at 1e-2, while their tensor-scale error stays at 7e-6 (r05:
block3's atrous layers left that group -- their zero-padded tiles run in sixteen products with constants {1, 1.5, 2}:
2e-4 element-wise, held to 1e-3 again).  The rule
that follows (include/monopsr_hip.h, mpsr_set_winograd_policy): MPSR_WINOGRAD_AUTO is the default -- the tensor-scale
error is what the path's 1e-3 budget is about, and end to end on heavy-tailed features (100x outliers, last test) the
decoder's small outputs are 1.2e-3 off where fp32 arithmetic alone, without any Winograd kernel, is 8.6e-4 off; a
caller that needs the last factor on every small output under such inputs selects MPSR_WINOGRAD_OFF, which this file
checks restores the direct kernels' 1e-4 per layer.

Kernels covered: Winograd F(4x4,3x3) (csrc/winograd4.hip: transform constants up to 8 and 1/24, the decoder's conv2_2 /
conv3_2), block3's conv2 on its atrous sub-grids (csrc/winograd3z.hip: sixteen products per zero-padded tile),
F(3x3,3x3) on tiles with halos (csrc/winograd3.hip: blocks 1-2, small-batch decoder), the upsampled-conv tap GEMM + gather
(csrc/upconv.hip, conv2_1 / conv3_1), the persistent pointwise kernel, and the whole decoder chain end to end.
"""
import numpy as np
import pytest
import torch

from oracle import net as onet

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def hostile_map(rng, shape, sparsity=0.9, outlier_frac=0.01, outlier_gain=1e3):
    """relu(N(0,1)) with `sparsity` of the entries zeroed and `outlier_frac` of them multiplied by `outlier_gain`."""
    x = np.abs(rng.standard_normal(shape)).astype(np.float32)
    x *= rng.random(shape) >= sparsity
    x *= np.where(rng.random(shape) < outlier_frac, outlier_gain, 1.0).astype(np.float32)
    return x.astype(np.float32)


def trained_like_filter(rng, kh, kw, cin, cout):
    """He-scaled taps times a log-normal per-output-channel gain (two decades: what folding BatchNorm's gamma / sigma
    into a trained layer produces), HWIO."""
    w = rng.standard_normal((kh, kw, cin, cout)) * np.sqrt(2.0 / (kh * kw * cin))
    gain = np.exp(rng.uniform(np.log(0.1), np.log(10.0), cout))
    return (w * gain).astype(np.float32)


def errors(got, ref):
    """(max|err| / max|ref|, max relative error over the elements with |ref| > 1e-3 max|ref|)"""
    got = got.detach().cpu().double().numpy() if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    ref = ref.detach().cpu().double().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref, np.float64)
    assert got.shape == ref.shape
    scale = np.abs(ref).max()
    big = np.abs(ref) > 1e-3 * scale
    return float(np.abs(got - ref).max() / scale), float((np.abs(got - ref)[big] / np.abs(ref)[big]).max())


def _conv_ref(x, w, bias, rate=1, relu=True):
    y = onet.tf_conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double(), rate=rate)
    y = y + torch.from_numpy(bias).double()
    return torch.relu(y) if relu else y


# (kernel, forcing knob value, B, H, W, C, N, dilation): the decoder's and block3's real channel counts
CASES = [
    ("winograd F(4x4,3x3)", 2, 8, 24, 24, 256, 256, 1),
    ("winograd F(4x4,3x3)", 2, 4, 48, 48, 128, 128, 1),
    ("sixteen-product zero-padded tiles (block3, atrous)", 3, 64, 12, 12, 256, 256, 4),
    ("winograd F(3x3,3x3) tiles with halos (block2)", 3, 64, 12, 12, 128, 128, 2),
    ("winograd F(3x3,3x3) tiles with halos (block1)", 3, 64, 12, 12, 64, 64, 1),
    ("winograd F(3x3,3x3) tiles with halos (decoder, small batch)", 3, 8, 24, 24, 256, 256, 1),
    ("direct implicit GEMM", 0, 8, 24, 24, 256, 256, 1),
]


@pytest.mark.parametrize("name,wino,B,H,Wd,C,N,dil", CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_conv3x3_kernels_on_heavy_tailed_maps(name, wino, B, H, Wd, C, N, dil, relu):
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(1000 * C + N + dil + int(relu))
    x = hostile_map(rng, (B, H, Wd, C))
    w = trained_like_filter(rng, 3, 3, C, N)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    ref = _conv_ref(x, w, bias, dil, relu)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_winograd(wino)
    try:
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias), None, 3, 3, dil, relu, split_k=0)
    finally:
        lib.mpsr_debug_set_conv_winograd(-1)
    tens, elem = errors(got, ref)
    print("%s relu=%d: tensor-scale %.2e, element-wise (|ref| > 1e-3 max) %.2e" % (name, relu, tens, elem))
    assert tens <= 1e-4, (name, tens)
    # History of this bound (r04 review: "do not move a tolerance again without recording the old one and why"): the test
    # was first written with 1e-3 for every kernel; the Winograd kernels measured 1.9e-3 .. 4.2e-3 on these maps
    # (profiles/r04_hostile_inputs.txt) and the bound for them was set to 1e-2 in r04.  r05 looked for a cheaper way out
    # than the policy switch -- other interpolation points for F(4x4,3x3) / F(3x3,3x3) simulated in float32 on these maps
    # (profiles/r05_winograd_points.txt): best case 0.6x of the current error, not the 4x needed -- so the bound stays, the
    # header / README say that the default misses an element-wise 1e-3 here by up to ~4x, and the configuration that
    # keeps it (MPSR_WINOGRAD_OFF, next test) is timed in the bench line (`winograd_off_mode`).  Later in r05 block3's
    # case ("winograd F(3x3,3x3) atrous" until then, 1.4e-3 .. 2e-3 measured, bound 1e-2) moved to the sixteen-product
    # kernel, measures 2.1e-4 .. 2.4e-4 (profiles/r05_hostile_inputs.txt) and is held to the original 1e-3 again.
    assert elem <= (1e-2 if name.startswith("winograd") else 1e-3), (name, elem)


@pytest.mark.parametrize("B,H,Wd,C,N,dil,kind", [(120, 24, 24, 256, 64, 1, 3), (64, 12, 12, 256, 256, 4, 4)])
def test_winograd_policy_off_restores_elementwise_accuracy(B, H, Wd, C, N, dil, kind):
    """Shapes the library sends to a Winograd kernel by itself (mpsr_conv2d_plan kind 3 / 4).  Under MPSR_WINOGRAD_OFF the
    plan says implicit GEMM and every element with |ref| > 1e-3 max is within 1e-3 of itself on the heavy-tailed map."""
    import ctypes
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B + C + dil)
    x = hostile_map(rng, (B, H, Wd, C))
    w = trained_like_filter(rng, 3, 3, C, N)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    ref = _conv_ref(x, w, bias, dil, True)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()

    def plan_kind():
        k, ex = ctypes.c_int(0), ctypes.c_double(0.0)
        _lib.check(lib.mpsr_conv2d_plan(B, H, Wd, C, N, 3, 3, dil, ctypes.byref(k), ctypes.byref(ex)))
        return k.value
    assert _lib.set_winograd_policy("auto") == "auto" and plan_kind() == kind
    auto = errors(dn.conv2d(_dev(x), _dev(w_ok), _dev(bias), None, 3, 3, dil, True, split_k=0), ref)
    try:
        _lib.set_winograd_policy("off")
        assert plan_kind() == 0
        off = errors(dn.conv2d(_dev(x), _dev(w_ok), _dev(bias), None, 3, 3, dil, True, split_k=0), ref)
    finally:
        _lib.set_winograd_policy("auto")
    print("plan kind %d: auto tensor %.2e element %.2e | off tensor %.2e element %.2e" % ((kind,) + auto + off))
    # (kind 4 at one tile per sub-grid is the sixteen-product kernel: 4.1e-4 measured, inside 1e-3 without the switch)
    assert auto[0] <= 1e-4 and auto[1] <= (1e-3 if kind == 4 else 1e-2)
    assert off[0] <= 1e-5 and off[1] <= 1e-3


@pytest.mark.parametrize("B,H,Wd,C,N,dil,auto_kind,acc_kind", [
    (128, 24, 24, 256, 256, 1, 3, 1),   # decoder conv2_2: F(4x4,3x3) -> F(2x2,3x3)
    (32, 48, 48, 128, 128, 1, 3, 1),    # decoder conv3_2: F(4x4,3x3) -> F(2x2,3x3)
    (64, 12, 12, 256, 256, 4, 4, 4),    # block3 conv2: the sixteen-product form under both
    (64, 12, 12, 128, 128, 2, 4, 0),    # block2 conv2: F(3x3,3x3) tiles with halos -> border-class implicit GEMM
    (64, 12, 12, 64, 64, 1, 4, 0),      # block1 conv2: F(3x3,3x3) tiles with halos -> implicit GEMM
])
def test_winograd_policy_accurate_keeps_an_elementwise_1e_3(B, H, Wd, C, N, dil, auto_kind, acc_kind):
    """MPSR_WINOGRAD_ACCURATE (r06; the r05 review's "actionable residue": F(4x4,3x3) and the halo tiles miss an
    element-wise 1e-3 on heavy-tailed maps by 2-4x under the default policy): the plan moves the decoder's dense layers
    to F(2x2,3x3) and blocks 1-2 to the direct kernels, keeps the sixteen-product form, and EVERY element with
    |ref| > 1e-3 max is within 1e-3 of itself on the heavy-tailed map; per call and process-wide."""
    import ctypes
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(B + C + 7 * dil)
    x = hostile_map(rng, (B, H, Wd, C))
    w = trained_like_filter(rng, 3, 3, C, N)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    ref = _conv_ref(x, w, bias, dil, True)
    w_ok, _ = W.fold_conv(w)
    lib = _lib.lib()

    def plan_kind():
        k, ex = ctypes.c_int(0), ctypes.c_double(0.0)
        _lib.check(lib.mpsr_conv2d_plan(B, H, Wd, C, N, 3, 3, dil, ctypes.byref(k), ctypes.byref(ex)))
        return k.value
    assert _lib.set_winograd_policy("auto") == "auto" and plan_kind() == auto_kind
    per_call = errors(dn.conv2d(_dev(x), _dev(w_ok), _dev(bias), None, 3, 3, dil, True, split_k=0,
                                winograd_policy="accurate"), ref)
    try:
        assert _lib.set_winograd_policy("accurate") == "auto"
        assert plan_kind() == acc_kind
        got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias), None, 3, 3, dil, True, split_k=0)
    finally:
        assert _lib.set_winograd_policy("auto") == "accurate"
    acc = errors(got, ref)
    print("accurate policy, plan kind %d -> %d: tensor %.2e element %.2e" % ((auto_kind, acc_kind) + acc))
    assert acc == per_call  # (the option of a call and the process-wide default pick the same kernel: same bits)
    assert acc[0] <= 1e-5 and acc[1] <= 1e-3, acc


@pytest.mark.parametrize("h,C,N", [(12, 512, 256), (24, 256, 128)])
def test_upsampled_conv_on_heavy_tailed_maps(h, C, N):
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(h + C)
    B = 4
    x = hostile_map(rng, (B, h, h, C))
    w = trained_like_filter(rng, 3, 3, C, N)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    up = onet.tf_resize_bilinear(torch.from_numpy(x).double(), 2 * h, 2 * h, True)
    ref = torch.relu(onet.tf_conv2d(up, torch.from_numpy(w).double()) + torch.from_numpy(bias).double())
    w_ok, _ = W.fold_conv(w)
    got = dn.conv3x3_upsampled(_dev(x), (2 * h, 2 * h), _dev(w_ok), _dev(bias), True, True)
    tens, elem = errors(got, ref)
    print("upsampled conv %dx%d: tensor-scale %.2e, element-wise %.2e" % (h, h, tens, elem))
    assert tens <= 1e-5 and elem <= 1e-3, (tens, elem)


def test_pointwise_with_residual_on_heavy_tailed_maps():
    from monopsr_amd.core import device_net as dn
    rng = np.random.default_rng(7)
    B, C, N = 8, 256, 1024
    x = hostile_map(rng, (B, 12, 12, C))
    res = hostile_map(rng, (B, 12, 12, N))
    w = trained_like_filter(rng, 1, 1, C, N)
    bias = (rng.standard_normal(N) * 0.1).astype(np.float32)
    ref = torch.relu(onet.tf_conv2d(torch.from_numpy(x).double(), torch.from_numpy(w).double()) +
                     torch.from_numpy(bias).double() + torch.from_numpy(res).double())
    w_ok = np.ascontiguousarray(w.reshape(C, N).T)
    got = dn.conv2d(_dev(x), _dev(w_ok), _dev(bias), _dev(res), 1, 1, 1, True, split_k=0)
    tens, elem = errors(got, ref)
    print("pointwise + residual: tensor-scale %.2e, element-wise %.2e" % (tens, elem))
    assert tens <= 1e-5 and elem <= 1e-3, (tens, elem)


@pytest.mark.parametrize("policy", ["auto", "off", "accurate"])
def test_decoder_chain_on_heavy_tailed_features(policy):
    """squash + map decoder + xyz head end to end (mpsr_squash_decoder_fwd) on heavy-tailed trunk features, with
    decoder weights rescaled like a trained checkpoint's (per-channel gains over two decades folded into every conv):
    features_for_box_3d, features_for_map and inst_xyz_map_local against oracle/net.py in float64."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    rng = np.random.default_rng(21)
    B = 128  # (large enough for the library to pick the channel-blocked F(4x4,3x3) chain under "auto")
    weights = W.synthetic_weights(seed=9)
    rescaled = 0
    for k in list(weights):
        if (k.startswith("squash/") or k.startswith("map_decoder/")) and k.endswith("/weights") and weights[k].ndim == 4:
            gain = np.exp(rng.uniform(np.log(0.3), np.log(3.0), weights[k].shape[3])).astype(np.float32)
            weights[k] = weights[k] * gain
            rescaled += 1
    assert rescaled == 5
    net = dn.DeviceNet(weights)
    crop = hostile_map(rng, (B, 12, 12, 1024), outlier_gain=1e2)
    full = hostile_map(rng, (B, 12, 12, 1024), outlier_gain=1e2)
    ref = onet.squash_decoder(torch.from_numpy(crop).double(), torch.from_numpy(full).double(), weights, 48, 48)
    from monopsr_amd import _lib
    try:
        _lib.set_winograd_policy(policy)
        fb, fm, xyz = net.squash_decoder(_dev(crop), _dev(full), (48, 48), want_feat_map=True)
    finally:
        _lib.set_winograd_policy("auto")
    for name, got, r in (("features_for_box_3d", fb, ref[0]), ("features_for_map", fm, ref[1]), ("inst_xyz_map_local", xyz, ref[2])):
        tens, elem = errors(got, r)
        print("decoder (%s) %s: tensor-scale %.2e, element-wise %.2e" % (policy, name, tens, elem))
        assert tens <= 1e-4, (name, tens)
        # (five fp32 layers deep the direct kernels alone measure 8.6e-4 here; with F(4x4,3x3) on conv2_2 / conv3_2 1.2e-3:
        # the "auto" bound was 1e-3 when the test was written, measured 1.23e-3 (profiles/r04_hostile_inputs.txt) and set
        # to 2e-3 in r04; see the note in test_conv3x3_kernels_on_heavy_tailed_maps)
        assert elem <= (2e-3 if policy == "auto" else 1e-3), (name, elem)
