"""Backward kernels of the instance path (wgrad GEMM, dgrad via the forward kernel, bias / ReLU / max-pool / resize
gradients, Adam) against PyTorch-CPU float64 autograd of the same (BatchNorm-folded) formulation.

Tolerance 1e-4 of the gradient tensor's scale (fp32 accumulation over up to 10^5 pixels; atomics reorder sums)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import net as onet

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _close(got, ref, tol, name):
    got = got.detach().cpu().double().numpy()
    ref = ref.detach().cpu().double().numpy()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    scale = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref).max() / scale
    assert err <= tol, "%s: max err / scale = %.3e" % (name, err)


def _ref_conv(x, w_ok, b, res, kh, kw, dil, relu):
    """folded layer in float64 on the CPU: w_ok (N, kh*kw*C)"""
    N = w_ok.shape[0]
    C = x.shape[-1]
    w = w_ok.reshape(N, kh, kw, C).permute(0, 3, 1, 2)
    y = F.conv2d(x.permute(0, 3, 1, 2), w, None, 1, ((kh - 1) * dil // 2, (kw - 1) * dil // 2), dil)
    y = y.permute(0, 2, 3, 1)
    if b is not None:
        y = y + b
    if res is not None:
        y = y + res
    return torch.relu(y) if relu else y


CASES = [
    # B, H, W, C, N, k, dil, bias, residual, relu
    (2, 12, 12, 64, 32, 1, 1, True, False, True),
    (3, 12, 12, 32, 128, 1, 1, True, True, True),
    (2, 12, 12, 32, 32, 3, 1, True, False, True),
    (2, 12, 12, 64, 64, 3, 4, True, False, True),
    (2, 12, 12, 32, 48, 3, 2, False, False, False),
    (1, 24, 24, 16, 3, 3, 1, True, False, False),   # xyz-like N = 3
    (5, 9, 7, 36, 20, 3, 1, True, True, True),      # ragged
    (37, 1, 1, 260, 100, 1, 1, True, False, True),  # FC
    (2, 20, 20, 160, 136, 3, 1, True, False, True),  # more than one 128-wide tile in n and c
]


@pytest.mark.parametrize("case", CASES)
def test_conv_layer_gradients(case):
    from monopsr_amd.core import autograd_ops as ops
    B, H, Wd, C, N, k, dil, has_b, has_res, relu = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
    w = (rng.standard_normal((N, k * k * C)) / np.sqrt(k * k * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32) if has_b else None
    res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
    up = rng.standard_normal((B, H, Wd, N)).astype(np.float32)
    # reference
    xr = torch.from_numpy(x).double().requires_grad_(True)
    wr = torch.from_numpy(w).double().requires_grad_(True)
    br = torch.from_numpy(b).double().requires_grad_(True) if has_b else None
    rr = torch.from_numpy(res).double().requires_grad_(True) if has_res else None
    yr = _ref_conv(xr, wr, br, rr, k, k, dil, relu)
    (yr * torch.from_numpy(up).double()).sum().backward()
    # HIP
    wt, dwt = _dev(w), torch.zeros((N, k * k * C), device="cuda")
    bt, dbt = (_dev(b), torch.zeros((N,), device="cuda")) if has_b else (None, None)
    layer = ops.LayerRef(wt, bt, dwt, dbt, C, N, k, k, dil, relu)
    xt = _dev(x).requires_grad_(True)
    rt = _dev(res).requires_grad_(True) if has_res else None
    y = ops.conv2d(xt, layer, rt)
    _close(y, yr, 2e-6, "forward")
    (y * _dev(up)).sum().backward()
    _close(xt.grad, xr.grad, 1e-5, "dx")
    _close(dwt, wr.grad, 1e-4, "dw")
    if has_b:
        _close(dbt, br.grad, 1e-4, "db")
    if has_res:
        _close(rt.grad, rr.grad, 1e-6, "dresidual")


@pytest.mark.parametrize("chunk", range(4))
def test_conv_layer_gradients_random(chunk):
    """Seeded random sweep of the layer backward (weight gradient kernel with its per-tap valid rectangles and LDS row
    table, bias gradient, data gradient through the scheduled forward kernels) on small ragged shapes."""
    from monopsr_amd.core import autograd_ops as ops
    rng = np.random.default_rng(4000 + chunk)
    for case in range(12):
        k = int(rng.choice([1, 3]))
        dil = int(rng.choice([1, 2, 4])) if k == 3 else 1
        B, H, Wd = int(rng.integers(1, 6)), int(rng.integers(1, 14)), int(rng.integers(1, 14))
        C, N = int(rng.choice([4, 16, 32, 36, 64, 132])), int(rng.choice([3, 4, 20, 32, 64, 130]))
        has_b, has_res, relu = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
        x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
        w = (rng.standard_normal((N, k * k * C)) / np.sqrt(k * k * C)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32) if has_b else None
        res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
        up = rng.standard_normal((B, H, Wd, N)).astype(np.float32)
        xr = torch.from_numpy(x).double().requires_grad_(True)
        wr = torch.from_numpy(w).double().requires_grad_(True)
        br = torch.from_numpy(b).double().requires_grad_(True) if has_b else None
        rr = torch.from_numpy(res).double().requires_grad_(True) if has_res else None
        yr = _ref_conv(xr, wr, br, rr, k, k, dil, relu)
        (yr * torch.from_numpy(up).double()).sum().backward()
        wt, dwt = _dev(w), torch.zeros((N, k * k * C), device="cuda")
        bt, dbt = (_dev(b), torch.zeros((N,), device="cuda")) if has_b else (None, None)
        layer = ops.LayerRef(wt, bt, dwt, dbt, C, N, k, k, dil, relu)
        xt = _dev(x).requires_grad_(True)
        rt = _dev(res).requires_grad_(True) if has_res else None
        y = ops.conv2d(xt, layer, rt)
        tag = "chunk %d case %d: B%d H%d W%d C%d N%d k%d d%d b%d res%d relu%d" % (chunk, case, B, H, Wd, C, N, k, dil,
                                                                           has_b, has_res, relu)
        _close(y, yr, 5e-6, tag + " forward")
        (y * _dev(up)).sum().backward()
        _close(xt.grad, xr.grad, 2e-5, tag + " dx")
        _close(dwt, wr.grad, 1e-4, tag + " dw")
        if has_b:
            _close(dbt, br.grad, 1e-4, tag + " db")


@pytest.mark.parametrize("shape,k,s,pad", [((2, 24, 24, 16), 3, 2, "SAME"), ((2, 12, 12, 8), 2, 2, "VALID"),
                                           ((1, 7, 9, 4), 3, 2, "SAME")])
def test_max_pool_gradient(shape, k, s, pad):
    from monopsr_amd.core import autograd_ops as ops
    rng = np.random.default_rng(5)
    x = rng.standard_normal(shape).astype(np.float32)
    xr = torch.from_numpy(x).double().requires_grad_(True)
    yr = onet.tf_max_pool(xr, k, s, pad)
    up = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    (yr * torch.from_numpy(up).double()).sum().backward()
    xt = _dev(x).requires_grad_(True)
    y = ops.max_pool(xt, k, s, pad)
    (y * _dev(up)).sum().backward()
    _close(xt.grad, xr.grad, 1e-6, "max_pool dx")


@pytest.mark.parametrize("shape,out", [((2, 12, 12, 16), (24, 24)), ((1, 24, 24, 8), (48, 48)), ((2, 5, 7, 4), (9, 4)),
                                       ((1, 3, 3, 3), (40, 40)), ((1, 2, 2, 4), (40, 40)), ((1, 1, 1, 4), (5, 5))])
def test_resize_bilinear_gradient(shape, out):
    from monopsr_amd.core import autograd_ops as ops
    rng = np.random.default_rng(6)
    x = rng.standard_normal(shape).astype(np.float32)
    xr = torch.from_numpy(x).double().requires_grad_(True)
    yr = onet.tf_resize_bilinear(xr, out[0], out[1], True)
    up = rng.standard_normal(tuple(yr.shape)).astype(np.float32)
    (yr * torch.from_numpy(up).double()).sum().backward()
    xt = _dev(x).requires_grad_(True)
    y = ops.resize_bilinear(xt, out, True)
    (y * _dev(up)).sum().backward()
    _close(xt.grad, xr.grad, 1e-5, "resize dx")


def test_adam_step_matches_tf_formula():
    from monopsr_amd import _lib
    rng = np.random.default_rng(7)
    n = 10007
    p = rng.standard_normal(n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    pt, mt, vt = _dev(p), _dev(m), _dev(v)
    lr, b1, b2, eps = 8e-5, 0.9, 0.999, 1e-8
    pr, mr, vr = p.astype(np.float64), m.astype(np.float64), v.astype(np.float64)
    for step in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        _lib.check(_lib.lib().mpsr_adam_step(pt.data_ptr(), _dev(g).data_ptr(), mt.data_ptr(), vt.data_ptr(), n, lr,
                                             b1, b2, eps, step, 1.0, _lib.stream()))
        mr = b1 * mr + (1 - b1) * g
        vr = b2 * vr + (1 - b2) * g.astype(np.float64) ** 2
        pr = pr - lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step) * mr / (np.sqrt(vr) + eps)
    np.testing.assert_allclose(pt.cpu().numpy(), pr, rtol=1e-5, atol=1e-7)


def _ref_network(params, layers, crops, full_feat, n_trunk):
    """float64 CPU forward of trunk + decoder + xyz in the folded parameterisation, from the flat parameter vector."""
    P = params.double()

    def layer(i):
        L = layers[i]
        n = L.cout * L.kh * L.kw * L.cin
        w = P[L._w_off:L._w_off + n].view(L.cout, -1)
        b = P[L._b_off:L._b_off + L.cout] if L._b_off >= 0 else None
        return L, w, b

    def conv(x, i, res=None):
        L, w, b = layer(i)
        return _ref_conv(x, w, b, res, L.kh, L.kw, L.dilation, L.relu)

    B = crops.shape[0]
    xp = F.pad(crops.permute(0, 3, 1, 2), (3, 3, 3, 3))
    cols = F.unfold(xp, 7, stride=2)  # (B, 3*49, L) ordered (c, ky, kx)
    Lroot, w0, b0 = layer(0)
    oh = (crops.shape[1] + 6 - 7) // 2 + 1
    cols = cols.view(B, 3, 49, -1).permute(0, 3, 2, 1).reshape(B, -1, 147)  # -> (ky*7+kx)*3 + c
    x = torch.relu(cols @ w0[:, :147].t() + b0).view(B, oh, oh, -1)
    x = onet.tf_max_pool(x, 3, 2, "SAME")
    li = 1
    for units in (3, 4, 23):
        for u in range(units):
            res = x
            if u == 0:
                res = conv(x, li)
                li += 1
            t = conv(conv(x, li), li + 1)
            x = conv(t, li + 2, res)
            li += 3
    d = n_trunk
    sq = conv(full_feat, d + 1, conv(x, d))
    y = onet.tf_resize_bilinear(sq, 24, 24, True)
    y = conv(conv(y, d + 2), d + 3)
    y = onet.tf_resize_bilinear(y, 48, 48, True)
    fm = conv(conv(y, d + 4), d + 5)
    return conv(fm, d + 6), onet.tf_max_pool(sq, 2, 2, "VALID")


def test_network_training_gradients_end_to_end():
    """Narrow (1/4-width) trunk + decoder + xyz head: loss = Chamfer(pred cloud, GT) + small L2 on the box
    features; every parameter gradient of the flat buffer vs float64 CPU autograd; then one Adam step."""
    from monopsr_amd.core import train_net
    from monopsr_amd.core import weights as W
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance
    B, div = 2, 4
    weights = W.synthetic_weights(seed=51, width_div=div, heads=False)
    net = train_net.TrainNet(weights, width_div=div, with_heads=False)
    rng = np.random.default_rng(52)
    crops = (rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)
    full = np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0).astype(np.float32)
    gt = rng.standard_normal((B, 300, 3)).astype(np.float32) * 30
    # HIP
    net.zero_grad()
    feat = net.trunk(_dev(crops))
    fb, fm, xyz = net.squash_decoder(feat, _dev(full))
    pred = xyz.reshape(B, -1, 3)[:, :256]
    d1, _, d2, _ = tf_nndistance.nn_distance(pred.contiguous(), _dev(gt))
    loss = (d1.sum() + d2.sum()) / B + 1e-3 * (fb * fb).sum()
    loss.backward()
    torch.cuda.synchronize()
    # reference
    for L, r in zip(net.layers, W.pack_trunk(weights, W.CROP_SCOPE, div)[1] + W.pack_decoder(weights, div)[1]):
        pass
    base = 0
    recs = []
    for blob, records in (W.pack_trunk(weights, W.CROP_SCOPE, div), W.pack_decoder(weights, div)):
        for r in records:
            recs.append((r["w_off"] + base, (r["b_off"] + base) if r["b_off"] >= 0 else -1))
        base += blob.size
    for L, (wo, bo) in zip(net.layers, recs):
        L._w_off, L._b_off = wo, bo
    P = net.params.detach().cpu().double().requires_grad_(True)
    xyz_r, fb_r = _ref_network(P, net.layers, torch.from_numpy(crops).double(), torch.from_numpy(full).double(),
                               net.n_trunk)
    _close(xyz, xyz_r, 1e-4, "xyz forward")
    pr = xyz_r.reshape(B, -1, 3)[:, :256]
    dist = ((pr[:, :, None, :] - torch.from_numpy(gt).double()[:, None, :, :]) ** 2).sum(-1)
    loss_r = (dist.min(2).values.sum() + dist.min(1).values.sum()) / B + 1e-3 * (fb_r * fb_r).sum()
    loss_r.backward()
    np.testing.assert_allclose(float(loss), float(loss_r), rtol=1e-4)
    g, gr = net.grads.cpu().double(), P.grad
    # per-layer comparison (layers differ in gradient scale by orders of magnitude)
    worst = 0.0
    for L in net.layers:
        n = L.cout * L.kh * L.kw * L.cin
        a, b_ = g[L._w_off:L._w_off + n], gr[L._w_off:L._w_off + n]
        if L.kh == 1 and L.cin == 160:  # root: compare the 147 real columns (padding columns get no gradient)
            a, b_ = a.view(L.cout, 160)[:, :147], b_.view(L.cout, 160)[:, :147]
        worst = max(worst, float((a - b_).abs().max() / (b_.abs().max() + 1e-30)))
        if L._b_off >= 0:
            a, b_ = g[L._b_off:L._b_off + L.cout], gr[L._b_off:L._b_off + L.cout]
            worst = max(worst, float((a - b_).abs().max() / (b_.abs().max() + 1e-30)))
    assert worst < 2e-3, "worst per-layer gradient error %.3e" % worst
    before = net.params.clone()
    net.adam_step(lr=1e-3)
    assert float((net.params - before).abs().max()) > 0


@pytest.mark.parametrize("B,h,w,C,OH,OW,N,bias_relu", [(2, 12, 12, 128, 24, 24, 128, True), (2, 6, 9, 192, 12, 18, 256, False),
                                                      (1, 24, 24, 256, 48, 48, 128, True), (2, 5, 7, 128, 11, 16, 128, True)])
def test_upsampled_conv_backward_vs_float64_autograd(B, h, w, C, OH, OW, N, bias_relu):
    """UpsampledConv2dFn (resize_bilinear(align_corners) -> 3x3 conv as one tap-GEMM operator, csrc/upconv.hip): data,
    weight and bias gradients against float64 autograd of oracle/net.py's two TF-1.8 operators."""
    from monopsr_amd.core import autograd_ops as ops
    from monopsr_amd.core import weights as W
    from oracle import net as onet
    rng = np.random.default_rng(B + h + C + N)
    x = rng.standard_normal((B, h, w, C)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, C, N)) / np.sqrt(9 * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    up = rng.standard_normal((B, OH, OW, N)).astype(np.float32)
    w_ok, _ = W.fold_conv(wt)
    wd, bd = _dev(w_ok), _dev(b)
    L = ops.LayerRef(wd, bd, torch.zeros_like(wd), torch.zeros_like(bd), C, N, 3, 3, 1, bias_relu)
    assert ops.upsampled_conv_applies((B, h, w, C), L, (OH, OW))
    xd = _dev(x).requires_grad_(True)
    y = ops.upsampled_conv2d(xd, L, (OH, OW), True)
    (y * _dev(up)).sum().backward()
    torch.cuda.synchronize()
    x64 = torch.from_numpy(x).double().requires_grad_(True)
    w64 = torch.from_numpy(wt).double().requires_grad_(True)
    b64 = torch.from_numpy(b).double().requires_grad_(True)
    ref = onet.tf_conv2d(onet.tf_resize_bilinear(x64, OH, OW, True), w64) + b64
    if bias_relu:
        ref = torch.relu(ref)
    (ref * torch.from_numpy(up).double()).sum().backward()
    _close(y, ref, 1e-5, "forward")
    _close(xd.grad, x64.grad, 2e-5, "data gradient")
    dw_ref = w64.grad.permute(3, 0, 1, 2).reshape(N, -1)  # HWIO -> (N, 9 C)
    _close(L.dw, dw_ref, 1e-4, "weight gradient")
    _close(L.db, b64.grad, 1e-4, "bias gradient")


def test_bottleneck_units_as_one_autograd_node_give_the_same_gradients():
    """TrainNet.trunk runs every bottleneck unit as ONE autograd node (autograd_ops.BottleneckFn: the two gradients of
    the unit's input meet in the epilogue of conv1's data-gradient convolution instead of in an elementwise launch).
    Same kernels, and a + b == b + a: every parameter gradient must equal the layer-by-layer graph's.  (wgrad slices
    meet in fp32 atomics whose order is free, so "equal" is 1e-5 of each layer's scale; the forward is bit-identical.)"""
    from monopsr_amd.core import train_net
    from monopsr_amd.core import weights as W
    B, div = 3, 4
    weights = W.synthetic_weights(seed=71, width_div=div, heads=False)
    net = train_net.TrainNet(weights, width_div=div, with_heads=False)
    rng = np.random.default_rng(72)
    crops = _dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32))
    out = {}
    for fused in (True, False):
        net.fused_units = fused
        net.zero_grad()
        feat = net.trunk(crops)
        (feat * feat).sum().backward()
        torch.cuda.synchronize()
        out[fused] = (feat.detach().clone(), net.grads.clone())
    net.fused_units = True
    assert torch.equal(out[True][0], out[False][0])
    g1, g0 = out[True][1], out[False][1]
    assert float(g0.abs().max()) > 0
    for n, L in enumerate(net.layers[:net.n_trunk]):  # L.dw is a view into net.grads: compare the same slice of both
        lo = (L.dw.data_ptr() - net.grads.data_ptr()) // 4
        a, b_ = g1[lo:lo + L.dw.numel()], g0[lo:lo + L.dw.numel()]
        assert float((a - b_).abs().max()) <= 1e-5 * float(b_.abs().max() + 1e-30), n


@pytest.mark.parametrize("M,K,N,res", [(256 * 144, 256, 1024, True), (5 * 96 + 17, 1024, 256, False),
                                       (3 * 144, 256, 512, True)])
def test_masked_pointwise_data_gradient_is_the_unmasked_one_times_the_relu_mask(M, K, N, res):
    """mpsr_relu_bitmask + mpsr_conv1x1_masked_f32 (the masked store path of the persistent pointwise kernel): kept
    elements are BIT-identical to the plain launch, the others exactly zero -- i.e. conv -> mpsr_relu_grad in one
    launch.  Ragged M (not a multiple of 32 or of the 96-row tile) included."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    lib = _lib.lib()
    rng = np.random.default_rng(M + K + N)
    x = _dev(rng.standard_normal((1, 1, M, K)).astype(np.float32))
    w = _dev((rng.standard_normal((N, K)) / np.sqrt(K)).astype(np.float32))
    r = _dev(rng.standard_normal((1, 1, M, N)).astype(np.float32)) if res else None
    act = _dev(np.maximum(rng.standard_normal((M, N)), 0).astype(np.float32))  # the post-ReLU tensor the mask is of
    assert lib.mpsr_conv1x1_masked_applies(M, K, N) == 1
    bits = torch.zeros((lib.mpsr_relu_bitmask_words(M, N),), dtype=torch.int32, device="cuda")
    _lib.check(lib.mpsr_relu_bitmask(_lib.ptr(act), M, N, _lib.ptr(bits), _lib.stream()))
    got = torch.full((M, N), 7.0, dtype=torch.float32, device="cuda")
    _lib.check(lib.mpsr_conv1x1_masked_f32(_lib.ptr(x), M, K, _lib.ptr(w), None, _lib.ptr(r), _lib.ptr(bits),
                                           _lib.ptr(got), N, _lib.stream()))
    lib.mpsr_debug_set_conv_pointwise(1)  # (few rows would otherwise go to the implicit GEMM: another summation order)
    try:
        plain = dn.conv2d(x, w, None, r, 1, 1, 1, False, split_k=0).reshape(M, N)
    finally:
        lib.mpsr_debug_set_conv_pointwise(-1)
    want = torch.where(act > 0, plain, torch.zeros_like(plain))
    assert torch.equal(got, want)
    # the bit layout the header documents: bit b <-> row 32 g + (b & 3) + 8 ((b >> 2) & 3) + 4 (b >> 4)
    b = bits.cpu().numpy().view(np.uint32).reshape(-1, N)
    a = act.cpu().numpy() > 0
    row_of = [(bb & 3) + 8 * ((bb >> 2) & 3) + 4 * (bb >> 4) for bb in range(32)]
    assert sorted(row_of) == list(range(32))
    for m in (0, 1, 4, 31, 32, 37, M - 1):
        bb = row_of.index(m & 31)
        assert np.array_equal((b[m >> 5] >> bb) & 1, a[m].astype(np.uint32)), m
    # ... and the forward launch that writes the words of its own result: y bit-identical to the plain launch, the
    # words equal to mpsr_relu_bitmask(y) on the rows that exist
    bias = _dev(rng.standard_normal(N).astype(np.float32))
    y2 = torch.empty((M, N), dtype=torch.float32, device="cuda")
    bits2 = torch.zeros_like(bits)
    _lib.check(lib.mpsr_conv1x1_relu_bitmask_f32(_lib.ptr(x), M, K, _lib.ptr(w), _lib.ptr(bias), _lib.ptr(r), 1,
                                                 _lib.ptr(y2), _lib.ptr(bits2), N, _lib.stream()))
    lib.mpsr_debug_set_conv_pointwise(1)
    try:
        y_plain = dn.conv2d(x, w, bias, r, 1, 1, 1, True, split_k=0).reshape(M, N)
    finally:
        lib.mpsr_debug_set_conv_pointwise(-1)
    assert torch.equal(y2, y_plain)
    bits3 = torch.zeros_like(bits)
    _lib.check(lib.mpsr_relu_bitmask(_lib.ptr(y_plain), M, N, _lib.ptr(bits3), _lib.stream()))
    b2 = bits2.cpu().numpy().view(np.uint32).reshape(-1, N)
    b3 = bits3.cpu().numpy().view(np.uint32).reshape(-1, N)
    valid = np.zeros(32, dtype=np.uint32)  # bits of the last word group whose rows exist
    for bb in range(32):
        valid[bb] = 1 if (M - 1) // 32 * 32 + row_of[bb] < M else 0
    vmask = np.uint32(sum(int(v) << i for i, v in enumerate(valid)))
    assert np.array_equal(b2[:-1], b3[:-1]) and np.array_equal(b2[-1] & vmask, b3[-1])
    assert 0.2 < float((y_plain > 0).float().mean()) < 0.8
    assert lib.mpsr_conv1x1_masked_applies(M, 128, N) == 0  # short K: the caller keeps conv + relu_grad
    rc = lib.mpsr_conv1x1_masked_f32(_lib.ptr(x), M, 128, _lib.ptr(w), None, None, _lib.ptr(bits), _lib.ptr(got), N,
                                     _lib.stream())
    assert rc != 0 and b"masked" in lib.mpsr_last_error()


@pytest.mark.parametrize("B,H,C,N,k,dil", [(64, 12, 256, 256, 3, 4), (3, 12, 64, 32, 3, 4), (2, 6, 32, 64, 3, 2),
                                            (2, 12, 64, 64, 3, 1), (2, 5, 128, 256, 1, 1)])
def test_conv2d_relu_masked_is_conv_then_relu_grad(B, H, C, N, k, dil):
    """mpsr_conv2d_relu_masked_f32 = the scheduled convolution followed by mpsr_relu_grad, bit for bit: the F(3x3,3x3)
    atrous kernel selects in its epilogue (first three shapes: 12x12 at dilation 4 and 6x6 at dilation 2 have 3x3
    sub-grids), the others run the two launches inside the call."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    lib = _lib.lib()
    rng = np.random.default_rng(B * 1000 + N)
    x = _dev(rng.standard_normal((B, H, H, C)).astype(np.float32))
    w = _dev((rng.standard_normal((N, k * k * C)) / np.sqrt(k * k * C)).astype(np.float32))
    act = _dev(np.maximum(rng.standard_normal((B, H, H, N)), 0).astype(np.float32))
    ws = torch.empty((lib.mpsr_conv2d_scratch_floats(B, H, H, N),), dtype=torch.float32, device="cuda")
    got = torch.full((B, H, H, N), 7.0, dtype=torch.float32, device="cuda")
    _lib.check(lib.mpsr_conv2d_relu_masked_f32(_lib.ptr(x), B, H, H, C, _lib.ptr(w), _lib.ptr(act), _lib.ptr(got), N, k, k,
                                               dil, _lib.ptr(ws), ws.numel(), _lib.stream()))
    # (the plain launch of a small batch is cut along K -- winograd3z.hip, SPLIT -- and adds in another order; the masked
    # launches of the training path never are: compare like with like)
    lib.mpsr_debug_set_wino3z_split(0)
    try:
        plain = dn.conv2d(x, w, None, None, k, k, dil, False, split_k=0)
    finally:
        lib.mpsr_debug_set_wino3z_split(-1)
    want = torch.where(act > 0, plain, torch.zeros_like(plain))
    assert torch.equal(got, want)
    assert 0.3 < float((got != 0).float().mean()) < 0.7


def test_linked_bottleneck_units_give_the_same_gradients():
    """TrainNet.trunk chains its bottleneck units (autograd_ops.UnitLink): a unit's input gradient leaves conv1's
    data-gradient launch already masked by the previous unit's ReLU, and that unit skips its elementwise pass.  Full
    width (the masked launch takes K >= 256), two crops: every parameter gradient and the image-side gradient must
    equal the unlinked graph's."""
    from monopsr_amd import _lib
    from monopsr_amd.core import autograd_ops, train_net
    from monopsr_amd.core import weights as W
    B = 2
    net = train_net.TrainNet(W.synthetic_weights(seed=81, heads=False), with_heads=False)
    rng = np.random.default_rng(82)
    crops = _dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32))
    taken, emitted = [], []
    orig = autograd_ops._data_grad_masked

    def counting(L, g, x, residual, bits=None):
        dx = orig(L, g, x, residual, bits)
        if residual is not None:  # conv1's data gradient + the shortcut's (conv3's goes through conv2's mask: no residual)
            taken.append(dx is not None)
            emitted.append(bits is not None)
        return dx
    autograd_ops._data_grad_masked = counting
    out = {}
    # (at two crops the library would send the unlinked graph's 1x1 layers to the implicit GEMM -- another summation
    # order than the pointwise kernel the mask-writing launch is; with the pointwise kernel wherever it applies both
    # graphs run the same launches and the forward must be bit-identical)
    _lib.lib().mpsr_debug_set_conv_pointwise(1)
    try:
        for linked in (True, False):
            net.linked_units = linked
            net.zero_grad()
            feat = net.trunk(crops)
            (feat * feat).sum().backward()
            torch.cuda.synchronize()
            out[linked] = (feat.detach().clone(), net.grads.clone())
    finally:
        _lib.lib().mpsr_debug_set_conv_pointwise(-1)
        autograd_ops._data_grad_masked = orig
        net.linked_units = True
    assert sum(taken) == 23, taken  # every unit of block3 (conv1 has 256 outputs: K = 256); blocks 1-2 are narrower
    assert sum(emitted) == 22, emitted  # ... and units 2..23 got the mask words from the previous unit's conv3 launch
    assert torch.equal(out[True][0], out[False][0])
    g1, g0 = out[True][1], out[False][1]
    for n, L in enumerate(net.layers[:net.n_trunk]):
        lo = (L.dw.data_ptr() - net.grads.data_ptr()) // 4
        a, b_ = g1[lo:lo + L.dw.numel()], g0[lo:lo + L.dw.numel()]
        assert float((a - b_).abs().max()) <= 1e-5 * float(b_.abs().max() + 1e-30), n  # (wgrad slices meet in atomics)


def test_instance_trainer_step_reduces_loss():
    """Full training step on a 1/4-width copy: heads included (method-by-method output builder over differentiable
    FC layers), the reference's configured loss set (monopsr_model.py:554-958), per-variable clip, Adam; the loss
    must go down on a fixed batch and every layer must have received a gradient."""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    B, div = 4, 4
    cfg = config_utils.default_config()
    net = train_net.TrainNet(W.synthetic_weights(seed=61, width_div=div), width_div=div)
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, lr=1e-4)
    rng = np.random.default_rng(62)
    y1, x1 = rng.uniform(0, 150, B), rng.uniform(0, 1000, B)
    boxes = np.stack([y1, x1, y1 + rng.uniform(20, 200, B), x1 + rng.uniform(20, 200, B)], 1).astype(np.float32)
    sample = dict(rgb_image_crops=_dev((rng.standard_normal((B, 48, 48, 3)) * 50).astype(np.float32)),
                  full_img_feature_crop=_dev(np.maximum(rng.standard_normal((B, 12, 12, 1024 // div)), 0)
                                             .astype(np.float32)),
                  boxes_2d=_dev(boxes),
                  cam_p=_dev(np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)),
                  est_view_angs=_dev(rng.uniform(-0.5, 0.5, B).astype(np.float32)),
                  class_indices=torch.ones((B, 1), dtype=torch.int32, device="cuda"),
                  mean_lwh=_dev(np.tile(np.array([[3.88, 1.63, 1.53]], np.float32), (B, 1))),
                  prop_cen_z_offset=torch.full((B,), 2.178, device="cuda"))
    sample.update(trainer.synthetic_ground_truth(sample, seed=63))
    losses = [float(tr.step(sample)) for _ in range(8)]
    assert np.isfinite(losses).all()
    assert min(losses[-3:]) < 0.9 * losses[0], losses
    assert set(tr.losses_dict) == {"inst_xyz_map_local", "lwh_offs", "alpha_bins", "alpha_regs", "cen_z_offs",
                                   "cen_y_offs", "proj_err", "inst_depth_map_global"}
    # one more backward without the optimizer: every layer's weight gradient is populated
    net.zero_grad()
    tr.loss(tr.forward(sample), sample)[1].backward()
    tr.reducer.finish()
    empty = [i for i, L in enumerate(net.layers) if float(L.dw.abs().max()) == 0.0]
    assert not empty, empty


def test_dgrad_bank_is_the_per_layer_pack_of_every_layer_bit_for_bit():
    """mpsr_conv2d_dgrad_pack_batch (one launch per step for all layers: narrow heads padded to 4 output channels,
    FC layers with padded K, 1x1 / 3x3 / 7x7 taps) against mpsr_conv2d_dgrad_pack layer by layer; the trainer's
    backward reads the bank only between refresh() and invalidate(), and a step with the bank gives the data
    gradients of a step without it (same kernels on the same bits)."""
    from monopsr_amd import _lib
    from monopsr_amd.core import autograd_ops as ops
    from monopsr_amd.core import train_net
    from monopsr_amd.core import weights as W
    net = train_net.TrainNet(W.synthetic_weights(seed=131, width_div=4), width_div=4)
    bank = net.dgrad_bank
    assert bank is not None and not bank.fresh
    net.params.normal_(generator=torch.Generator(device="cuda").manual_seed(132))
    bank.flat.fill_(float("nan"))
    bank.refresh()
    assert bank.fresh
    lib = _lib.lib()
    narrow = 0
    assert bank.skipped == [0]  # the root convolution reads the image: no data gradient, no slice
    assert not hasattr(net.layers[0], "wd")
    for li, L in enumerate(net.layers):
        if li in bank.skipped:
            continue
        pad = (-L.cout) % 4
        narrow += pad > 0
        n4 = L.cout + pad
        w4 = F.pad(L.w, (0, 0, 0, pad)).contiguous() if pad else L.w
        ref = torch.empty((L.cin, L.kh * L.kw * n4), dtype=torch.float32, device="cuda")
        _lib.check(lib.mpsr_conv2d_dgrad_pack(_lib.ptr(w4), n4, L.kh, L.kw, L.cin, _lib.ptr(ref), _lib.stream()))
        assert L.wd.shape == ref.shape and torch.equal(L.wd, ref), (L.cin, L.cout, L.kh)
        assert ops._dgrad_filter(L, L.cin, n4, pad, ref.device) is L.wd
    assert narrow >= 1  # (the 3-channel xyz head)
    bank.invalidate()
    L = net.layers[5]
    assert ops._dgrad_filter(L, L.cin, L.cout, 0, torch.device("cuda")) is not L.wd
    # a stale bank is never read: change the weights, run a backward outside the trainer, compare with a fresh pack
    net.params.mul_(2.0)
    g = torch.randn((2, 12, 12, L.cout), device="cuda")
    dx = ops._data_grad(L, g, L.cin)
    bank.refresh()
    dx2 = ops._data_grad(L, g, L.cin)
    bank.invalidate()
    assert torch.equal(dx, dx2)


@pytest.mark.parametrize("B,C,N", [(256, 18432, 1024), (8, 4608, 256), (6, 4608, 256), (4, 4100, 27)])
def test_wide_fc_data_gradient_vs_float64(B, C, N):
    """dX of a few-row FC layer with a wide input (img_fc): autograd_ops._data_grad runs it transposed (rows = input
    features) when B % 4 == 0; every form against g @ w in float64."""
    from monopsr_amd.core import autograd_ops as ops
    gen = torch.Generator(device="cuda").manual_seed(B + C)
    w = torch.randn((N, C), device="cuda", generator=gen) / np.sqrt(N)
    g = torch.randn((B, 1, 1, N), device="cuda", generator=gen)
    L = ops.LayerRef(w, None, None, None, C, N, 1, 1, 1, False)
    dx = ops._data_grad(L, g, C)
    assert dx.shape == (B, 1, 1, C) and dx.is_contiguous()
    ref = g.view(B, N).double() @ w.double()
    err = float((dx.view(B, C).double() - ref).abs().max() / ref.abs().max())
    assert err < 1e-5, err


@pytest.mark.parametrize("B,H,W,C", [(2, 48, 48, 128), (3, 12, 12, 64), (1, 7, 10, 256), (5, 24, 24, 128), (1, 62, 62, 64)])
def test_thin_side_gradients_of_the_xyz_head_vs_float64(B, H, W, C):
    """csrc/thin_conv.hip: weight gradient (4 x 9 x C, + bias gradient) and data gradient (4 -> C channels) of a 3x3
    layer with four output channels -- the xyz-map head padded to 4 -- against float64 autograd, and against the general
    kernels (mpsr_debug_set_thin_conv(0)) on the same inputs; ragged strips (H not a multiple of 6), several images per
    wave, dw accumulating into what is already there."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    lib = _lib.lib()
    gen = torch.Generator(device="cuda").manual_seed(1000 * B + H + C)
    x = torch.randn((B, H, W, C), device="cuda", generator=gen)
    dy = torch.randn((B, H, W, 4), device="cuda", generator=gen)
    dy[..., 3] = 0  # (the padded channel of the real head; the kernels do not rely on it)
    w = torch.randn((4, 9 * C), device="cuda", generator=gen) / np.sqrt(9 * C)
    xd = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wdd = w.double().view(4, 3, 3, C).permute(0, 3, 1, 2).requires_grad_(True)
    F.conv2d(xd, wdd, padding=1).backward(dy.double().permute(0, 3, 1, 2))
    dw_ref = wdd.grad.permute(0, 2, 3, 1).reshape(4, 9 * C)
    dx_ref = xd.grad.permute(0, 2, 3, 1)
    db_ref = dy.double().sum((0, 1, 2))
    out = {}
    for thin in (1, 0):
        lib.mpsr_debug_set_thin_conv(thin)
        try:
            dw = torch.full((4, 9 * C), 0.5, device="cuda")
            db = torch.full((4,), -2.0, device="cuda")
            _lib.check(lib.mpsr_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), B, H, W, C, 4, 3, 3, 1, _lib.ptr(dw),
                                                 _lib.ptr(db), _lib.stream()))
            wd = torch.empty((C, 9 * 4), device="cuda")
            _lib.check(lib.mpsr_conv2d_dgrad_pack(_lib.ptr(w), 4, 3, 3, C, _lib.ptr(wd), _lib.stream()))
            dx = dn.conv2d(dy, wd, None, None, 3, 3, 1, False)
        finally:
            lib.mpsr_debug_set_thin_conv(1)
        out[thin] = (dw - 0.5, db + 2.0, dx)
        for got, ref, name in ((dw - 0.5, dw_ref, "dw"), (db + 2.0, db_ref, "db"), (dx, dx_ref, "dx")):
            err = float((got.double() - ref).abs().max() / ref.abs().max())
            assert err < 2e-5, (thin, name, err)
    assert float((out[1][2] - out[0][2]).abs().max()) < 1e-4 * float(out[0][2].abs().max())


def test_clip_by_norm_segments_matches_per_tensor_clip():
    """mpsr_clip_by_norm_segments over the trainer's chunk table == tf.clip_by_norm tensor by tensor."""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    cfg = config_utils.default_config()
    net = train_net.TrainNet(W.synthetic_weights(seed=91, width_div=4), width_div=4)
    tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, clip_norm=1.0)
    g = torch.Generator(device="cuda").manual_seed(5)
    net.grads.copy_(torch.randn(net.grads.shape, device="cuda", generator=g) * 0.02)
    net.layers[3].dw.mul_(1e-3)                      # a variable below the threshold stays untouched
    before = net.grads.clone()
    want = before.clone()
    base = net.grads.data_ptr()
    n_clipped = 0
    for L in net.layers:
        for t in (L.dw, L.db):
            if t is None:
                continue
            lo = (t.data_ptr() - base) // 4
            seg = want[lo:lo + t.numel()]
            norm = torch.linalg.vector_norm(seg.double())
            if norm > 1.0:
                seg.mul_((1.0 / norm).float())
                n_clipped += 1
    tr.clip_per_variable()
    assert n_clipped > 10
    assert float((net.grads - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9
    lo = (net.layers[3].dw.data_ptr() - base) // 4
    assert torch.equal(net.grads[lo:lo + net.layers[3].dw.numel()], before[lo:lo + net.layers[3].dw.numel()])
    # the norms are summed in a fixed order (r06: replicas of a data-parallel run must clip the same reduced gradient by
    # the same bits): the same gradients give the same bits run after run, variables spread over many chunks included
    seg, begin, length, sumsq, nseg = tr._clip
    assert int(torch.bincount(seg.long()).max()) > 8 and sumsq.numel() == nseg + seg.numel()
    runs = []
    for _ in range(6):
        net.grads.copy_(before)
        sumsq.fill_(float("nan"))
        tr.clip_per_variable()
        runs.append((sumsq[:nseg].clone(), net.grads.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    want_sq = torch.stack([before[b:b + n].double().square().sum() for b, n in zip(begin.tolist(), length.tolist())])
    want_sq = torch.zeros(nseg, dtype=torch.float64, device="cuda").index_add_(0, seg.long(), want_sq)
    assert float(((runs[0][0].double() - want_sq).abs() / want_sq).max()) < 1e-5
    # a scratch without room for the chunks' partial sums is refused
    from monopsr_amd import _lib
    rc = _lib.lib().mpsr_clip_by_norm_segments(_lib.ptr(net.grads), _lib.ptr(seg), _lib.ptr(begin), _lib.ptr(length),
                                               seg.numel(), _lib.ptr(sumsq), nseg, nseg, 1.0, _lib.stream())
    assert rc != 0 and b"n_segments + n_chunks" in _lib.lib().mpsr_last_error()


def test_fused_clip_adam_ema_equals_the_three_passes():
    """mpsr_clip_adam_ema_step (r06: squared norms, then ONE pass for clip -> Adam -> moving average; the gradient is not
    written back) against mpsr_clip_by_norm_segments -> mpsr_adam_step -> shadow.lerp_ on the same gradients, over four
    steps with a decaying learning rate: parameters, both Adam moments and the moving average agree to 1e-6 of their
    scale (the same arithmetic up to fused multiply-add contraction), variables under the clip threshold included, and
    the gradient buffer of the fused path is left as it came."""
    from monopsr_amd.core import config_utils, train_net, trainer
    from monopsr_amd.core import weights as W
    cfg = config_utils.default_config()
    opt = cfg.train_config.optimizer.adam_optimizer
    opt.learning_rate_type, opt.initial_learning_rate = 'exponential_decay', 1e-3
    opt.decay_steps, opt.decay_factor, opt.staircase = 2, 0.5, True
    opt.use_moving_average, opt.moving_average_decay = True, 0.9
    weights = W.synthetic_weights(seed=92, width_div=4)
    trs = []
    for fused in (False, True):
        net = train_net.TrainNet(weights, width_div=4)
        tr = trainer.InstanceTrainer(net, cfg.model_config, cfg.dataset_config, cfg.train_config, clip_norm=1.0)
        tr.fused_update = fused
        trs.append(tr)
    g = torch.Generator(device="cuda").manual_seed(6)
    # the variables' elements (the flat buffer also holds alignment padding, whose gradient is always zero in a real
    # step: the whole-buffer Adam pass leaves it alone for that reason, the chunk pass because no chunk covers it)
    seg, begin, length, _, _ = trs[1]._clip_table()
    covered = torch.zeros(trs[0].net.grads.shape, dtype=torch.bool, device="cuda")
    for lo, n in zip(begin.tolist(), length.tolist()):
        covered[lo:lo + n] = True
    assert 0.9 < float(covered.float().mean()) < 1.0
    for step in range(4):
        grads = torch.randn(trs[0].net.grads.shape, device="cuda", generator=g) * 0.02 * covered
        for tr in trs:
            tr.net.grads.copy_(grads)
            tr.net.layers[3].dw.mul_(1e-3)  # a variable below the threshold: not scaled
            before = tr.net.grads.clone()
            if tr.fused_update:
                if tr._clip is None:
                    tr._clip = tr._clip_table()
                tr.optimizer.apply_clipped_gradients(tr.net, tr.global_step, tr._clip, tr.clip_norm)
                assert torch.equal(tr.net.grads, before)
            else:
                tr.clip_per_variable()
                tr.optimizer.apply_gradients(tr.net, tr.global_step)
                assert not torch.equal(tr.net.grads, before)
            tr.global_step += 1
        a, b = trs
        assert a.net.step_count == b.net.step_count == step + 1
        for name, x, y in (("params", a.net.params, b.net.params), ("adam_m", a.net.adam_m, b.net.adam_m),
                           ("adam_v", a.net.adam_v, b.net.adam_v), ("average", a.optimizer.shadow, b.optimizer.shadow)):
            assert float((x - y).abs().max()) <= 1e-6 * float(x.abs().max()), (step, name)
    assert float((trs[0].net.params - trs[0].optimizer.shadow).abs().max()) > 0  # (the average really lags)
    # no clipping, no average: plain Adam over the chunks == mpsr_adam_step over the whole buffer
    nets = [train_net.TrainNet(weights, width_div=4) for _ in range(2)]
    tab = trs[1]._clip
    for net in nets:
        net.grads.copy_(grads)
    nets[0].adam_step(lr=1e-3)
    nets[1].clip_adam_ema_step(tab, 0.0, lr=1e-3)
    assert float((nets[0].params - nets[1].params).abs().max()) <= 1e-6 * float(nets[0].params.abs().max())


def test_wgrad_winograd_domain_vs_direct_and_fp64():
    """The Winograd F(4x4,3x3)-domain weight gradient of the decoder's dense 3x3 layers (csrc/winograd4_wgrad.hip:
    dU = sum over tiles of (A dY A^T) (.) (B^T d B), dg = G^T dU G) against the direct kernel on a layer-sized problem
    and against float64 autograd on a slice of its filters."""
    import ctypes
    from monopsr_amd import _lib
    lib = _lib.lib()
    B, H, W, C, N = 64, 48, 48, 128, 128
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn((B, H, W, C), device="cuda", generator=g).clamp_(min=0)
    dy = torch.randn((B, H, W, N), device="cuda", generator=g)
    nws = lib.mpsr_conv2d_wgrad_scratch_floats(B, H, W, C, N, 3, 3, 1)
    assert nws == 36 * N * C
    assert lib.mpsr_conv2d_wgrad_scratch_floats(B, H, W, C, N, 3, 3, 2) == 0      # atrous: direct kernel
    assert lib.mpsr_conv2d_wgrad_scratch_floats(B, H + 2, W, C, N, 3, 3, 1) == 0  # not 4x4 blocks
    ws = torch.empty((nws,), device="cuda")
    outs = []
    for wino in (0, 1):
        dw = torch.zeros((N, 9 * C), device="cuda")
        db = torch.zeros((N,), device="cuda")
        lib.mpsr_debug_set_wgrad_winograd(wino)
        try:
            _lib.check(lib.mpsr_conv2d_wgrad_ws_f32(x.data_ptr(), dy.data_ptr(), B, H, W, C, N, 3, 3, 1, dw.data_ptr(),
                                                    db.data_ptr(), ws.data_ptr(), nws, _lib.stream()))
        finally:
            lib.mpsr_debug_set_wgrad_winograd(1)
        outs.append((dw, db))
    scale = outs[0][0].abs().max().item()
    assert (outs[1][0] - outs[0][0]).abs().max().item() <= 1e-4 * scale
    assert not torch.equal(outs[1][0], outs[0][0])  # it really is the other evaluation
    assert (outs[1][1] - outs[0][1]).abs().max().item() <= 1e-4 * outs[0][1].abs().max().item()
    # float64 autograd for the first 4 filters
    xd = x.double().permute(0, 3, 1, 2)
    w = torch.zeros((4, C, 3, 3), dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv2d(xd, w, padding=1)
    (y * dy[..., :4].double().permute(0, 3, 1, 2)).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(4, 9 * C)
    assert (outs[1][0][:4].double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()
    # scratch too small or absent: the direct kernel, same result as mpsr_conv2d_wgrad_f32
    dw2 = torch.zeros((N, 9 * C), device="cuda")
    _lib.check(lib.mpsr_conv2d_wgrad_ws_f32(x.data_ptr(), dy.data_ptr(), B, H, W, C, N, 3, 3, 1, dw2.data_ptr(), None,
                                            None, 0, _lib.stream()))
    assert (dw2 - outs[0][0]).abs().max().item() <= 1e-5 * scale


@pytest.mark.parametrize("B,dil,C,N", [(128, 4, 128, 192), (136, 4, 256, 256), (600, 2, 128, 128)])
def test_wgrad_winograd3_domain_vs_direct_and_fp64(B, dil, C, N):
    """The Winograd F(3x3,3x3)-domain weight gradient of the atrous 3x3 layers whose pixel sub-grids are single tiles
    (csrc/winograd3_wgrad.hip: block3's conv2, 12x12 at dilation 4 -- dU = sum over tiles of (A dY A^T) (.) (B'^T d B'),
    dg = G'^T dU G'; one 400-accumulator wave per 32 x 32 block) against the border-class direct kernel on layer-sized
    problems (ragged tile counts: the last slice's steps run past the last tile) and against float64 autograd on a slice
    of the filters."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    H = W = 3 * dil
    g = torch.Generator(device="cuda").manual_seed(B + C)
    x = torch.randn((B, H, W, C), device="cuda", generator=g).clamp_(min=0)
    dy = torch.randn((B, H, W, N), device="cuda", generator=g)
    dy = dy * (torch.rand((B, H, W, N), device="cuda", generator=g) > 0.5)  # (a ReLU-masked gradient, as in the step)
    nws = 64  # (this form needs no scratch: it folds dU back in registers and accumulates into dw)
    assert lib.mpsr_conv2d_wgrad_scratch_floats(B, H, W, C, N, 3, 3, dil) == 0
    ws = torch.empty((nws,), device="cuda")
    outs = []
    for wino in (0, 1):
        dw = torch.full((N, 9 * C), 0.25, device="cuda")  # (the kernels ACCUMULATE into dw)
        db = torch.zeros((N,), device="cuda")
        lib.mpsr_debug_set_wgrad_winograd(wino)
        try:
            _lib.check(lib.mpsr_conv2d_wgrad_ws_f32(x.data_ptr(), dy.data_ptr(), B, H, W, C, N, 3, 3, dil, dw.data_ptr(),
                                                    db.data_ptr(), ws.data_ptr(), nws, _lib.stream()))
        finally:
            lib.mpsr_debug_set_wgrad_winograd(1)
        outs.append((dw - 0.25, db))
    scale = outs[0][0].abs().max().item()
    assert (outs[1][0] - outs[0][0]).abs().max().item() <= 1e-4 * scale
    assert not torch.equal(outs[1][0], outs[0][0])  # it really is the other evaluation
    assert (outs[1][1] - outs[0][1]).abs().max().item() <= 1e-4 * outs[0][1].abs().max().item()
    xd = x.double().permute(0, 3, 1, 2)
    w = torch.zeros((4, C, 3, 3), dtype=torch.float64, device="cuda", requires_grad=True)
    y = F.conv2d(xd, w, padding=dil, dilation=dil)
    (y * dy[..., 7:11].double().permute(0, 3, 1, 2)).sum().backward()
    ref = w.grad.permute(0, 2, 3, 1).reshape(4, 9 * C)
    assert (outs[1][0][7:11].double() - ref).abs().max().item() <= 1e-4 * ref.abs().max().item()


def test_batch_norm_finalize_kernels_vs_float64():
    """mpsr_batch_norm_finalize / _grad_finalize (r06: one launch each for the per-channel arithmetic between the passes)
    against the same expressions in torch float64, at a decay that makes the moving statistics' update visible; NULL
    moving statistics / NULL beta gradient leave those alone."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(17)
    M, C, eps, decay = 4097, 260, 1e-3, 0.5
    z = torch.randn((M, C), device="cuda", generator=g) * 3 + 7
    sums = torch.empty((2, C), dtype=torch.float64, device="cuda")
    _lib.check(lib.mpsr_batch_norm_stats(_lib.ptr(z), M, C, sums[0].data_ptr(), sums[1].data_ptr(), _lib.stream()))
    mm0 = torch.randn(C, device="cuda", generator=g)
    mv0 = torch.rand(C, device="cuda", generator=g) + 0.5
    mm, mv = mm0.clone(), mv0.clone()
    out = torch.empty((2, C), device="cuda")
    _lib.check(lib.mpsr_batch_norm_finalize(sums[0].data_ptr(), sums[1].data_ptr(), _lib.ptr(z), M, C, eps, decay,
                                            _lib.ptr(mm), _lib.ptr(mv), out[0].data_ptr(), out[1].data_ptr(),
                                            _lib.stream()))
    z64 = z.double()
    mean, var = z64.mean(0), z64.var(0, unbiased=False)
    rel = lambda a, b: float((a.double() - b).abs().max() / b.abs().max())
    assert rel(out[0], mean) < 1e-6
    assert rel(out[1], torch.rsqrt(var + eps)) < 1e-6
    assert rel(mm, mm0.double() * decay + (1 - decay) * mean) < 1e-6
    assert rel(mv, mv0.double() * decay + (1 - decay) * var * M / (M - 1)) < 1e-6
    out2 = torch.empty((2, C), device="cuda")
    _lib.check(lib.mpsr_batch_norm_finalize(sums[0].data_ptr(), sums[1].data_ptr(), _lib.ptr(z), M, C, eps, decay,
                                            None, None, out2[0].data_ptr(), out2[1].data_ptr(), _lib.stream()))
    assert torch.equal(out, out2)
    # backward side
    sg = torch.randn((2, C), dtype=torch.float64, device="cuda", generator=g) * 100
    db0 = torch.randn(C, device="cuda", generator=g)
    db = db0.clone()
    means = torch.empty((2, C), device="cuda")
    _lib.check(lib.mpsr_batch_norm_grad_finalize(sg[0].data_ptr(), sg[1].data_ptr(), float(M), C, _lib.ptr(db),
                                                 means[0].data_ptr(), means[1].data_ptr(), _lib.stream()))
    assert torch.equal(db, db0 + sg[0].float())
    assert rel(means, sg / M) < 2e-7
    assert lib.mpsr_batch_norm_grad_finalize(sg[0].data_ptr(), sg[1].data_ptr(), 0.0, C, None, means[0].data_ptr(),
                                             means[1].data_ptr(), _lib.stream()) != 0


@pytest.mark.parametrize("shape", [(3, 5, 7, 132, 68), (2, 12, 12, 256, 1024), (64, 12, 12, 1024, 256), (5, 1, 1, 520, 36),
                                   (1, 3, 3, 4, 4)])
def test_pointwise_wgrad_direct_kernel_vs_float64_and_the_lds_kernel(shape):
    """pw_wgrad_direct_kernel (r06: 1x1 weight gradients with the operands loaded straight into the MFMA's source
    registers, no LDS) against float64 and against conv_wgrad_kernel on the same inputs: ragged tiles (N, C not multiples
    of 128 / 64 / 32), pixel counts that are not multiples of the 32-row step or of the ring depth, slices shorter than the
    ring, weight AND bias gradient, accumulation into a non-zero buffer."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    B, H, Wd, C, N = shape
    g = torch.Generator(device="cuda").manual_seed(B * 131 + C)
    x = torch.randn((B, H, Wd, C), device="cuda", generator=g)
    dy = torch.randn((B, H, Wd, N), device="cuda", generator=g)
    dw0 = torch.randn((N, C), device="cuda", generator=g)
    db0 = torch.randn((N,), device="cuda", generator=g)
    want_w = dw0.double() + dy.reshape(-1, N).double().t() @ x.reshape(-1, C).double()
    want_b = db0.double() + dy.reshape(-1, N).double().sum(0)
    got = {}
    try:
        for direct in (1, 0):
            lib.mpsr_debug_set_wgrad_direct(direct)
            dw, db = dw0.clone(), db0.clone()
            _lib.check(lib.mpsr_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), B, H, Wd, C, N, 1, 1, 1, _lib.ptr(dw),
                                                 _lib.ptr(db), _lib.stream()))
            got[direct] = (dw, db)
            scale_w, scale_b = float(want_w.abs().max()), float(want_b.abs().max())
            assert float((dw.double() - want_w).abs().max()) < 2e-6 * scale_w * max(1.0, (B * H * Wd) ** 0.5 / 16), direct
            assert float((db.double() - want_b).abs().max()) < 2e-6 * scale_b * max(1.0, (B * H * Wd) ** 0.5 / 16), direct
    finally:
        lib.mpsr_debug_set_wgrad_direct(0)
    assert float((got[1][0] - got[0][0]).abs().max()) < 1e-5 * float(want_w.abs().max())


@pytest.mark.parametrize("M,C", [(4097, 260), (256 * 24 * 24, 256), (33, 8)])
def test_batch_norm_backward_with_the_mask_rebuilt_from_z_is_bit_identical(M, C):
    """mpsr_batch_norm_grad_sums_z / _grad_z (r06: the ReLU mask recomputed from z with the forward's own fused
    multiply-add, y not read) == the passes that read y, bit for bit, on the y mpsr_batch_norm_apply produced -- including
    channels whose pre-activation sits at the rounding edge of zero."""
    from monopsr_amd import _lib
    lib = _lib.lib()
    g = torch.Generator(device="cuda").manual_seed(M + C)
    z = torch.randn((M, C), device="cuda", generator=g) * 2 + 1
    dy = torch.randn((M, C), device="cuda", generator=g)
    sums = torch.empty((2, C), dtype=torch.float64, device="cuda")
    _lib.check(lib.mpsr_batch_norm_stats(_lib.ptr(z), M, C, sums[0].data_ptr(), sums[1].data_ptr(), _lib.stream()))
    st = torch.empty((2, C), device="cuda")
    _lib.check(lib.mpsr_batch_norm_finalize(sums[0].data_ptr(), sums[1].data_ptr(), _lib.ptr(z), M, C, 1e-3, 0.999, None,
                                            None, st[0].data_ptr(), st[1].data_ptr(), _lib.stream()))
    beta = torch.randn(C, device="cuda", generator=g) * 0.5
    # put many elements exactly AT the edge: z = mean - beta / inv_std gives a pre-activation of ~0
    edge = (st[0] - beta / st[1])
    z[::7] = edge
    z[3::11] = edge * (1 + 1e-7)
    y = torch.empty_like(z)
    _lib.check(lib.mpsr_batch_norm_apply(_lib.ptr(z), M, C, st[0].data_ptr(), st[1].data_ptr(), _lib.ptr(beta), 1,
                                         _lib.ptr(y), _lib.stream()))
    assert 0.2 < float((y > 0).float().mean()) < 0.9
    out = {}
    for mode in ("y", "z"):
        s2 = torch.empty((2, C), dtype=torch.float64, device="cuda")
        if mode == "y":
            _lib.check(lib.mpsr_batch_norm_grad_sums(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(z), M, C, st[0].data_ptr(),
                                                     st[1].data_ptr(), s2[0].data_ptr(), s2[1].data_ptr(), _lib.stream()))
        else:
            _lib.check(lib.mpsr_batch_norm_grad_sums_z(_lib.ptr(dy), _lib.ptr(z), M, C, st[0].data_ptr(), st[1].data_ptr(),
                                                       _lib.ptr(beta), s2[0].data_ptr(), s2[1].data_ptr(), _lib.stream()))
        means = (s2 / M).float().contiguous()
        dz = torch.empty_like(z)
        if mode == "y":
            _lib.check(lib.mpsr_batch_norm_grad(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(z), M, C, st[0].data_ptr(),
                                                st[1].data_ptr(), means[0].data_ptr(), means[1].data_ptr(), _lib.ptr(dz),
                                                _lib.stream()))
        else:
            _lib.check(lib.mpsr_batch_norm_grad_z(_lib.ptr(dy), _lib.ptr(z), M, C, st[0].data_ptr(), st[1].data_ptr(),
                                                  _lib.ptr(beta), means[0].data_ptr(), means[1].data_ptr(), _lib.ptr(dz),
                                                  _lib.stream()))
        out[mode] = (s2, dz)
    # the masks are the same bits; the fp64 column sums are combined with atomics (order), so they agree to rounding
    assert float(((out["y"][0] - out["z"][0]).abs() / (out["y"][0].abs() + 1e-9)).max()) < 1e-12
    mask = (y > 0)
    gz = torch.where(mask, dy, torch.zeros_like(dy))
    # dz elementwise: recompute both from ONE set of means to compare the masks exactly
    means = (out["y"][0] / M).float().contiguous()
    dzs = []
    for fn, args in ((lib.mpsr_batch_norm_grad, (_lib.ptr(dy), _lib.ptr(y), _lib.ptr(z))),
                     (lib.mpsr_batch_norm_grad_z, (_lib.ptr(dy), _lib.ptr(z)))):
        dz = torch.empty_like(z)
        tail = (st[0].data_ptr(), st[1].data_ptr()) + ((_lib.ptr(beta),) if fn is lib.mpsr_batch_norm_grad_z else ()) + \
            (means[0].data_ptr(), means[1].data_ptr(), _lib.ptr(dz), _lib.stream())
        _lib.check(fn(*args, M, C, *tail))
        dzs.append(dz)
    assert torch.equal(dzs[0], dzs[1])
    assert float(gz.abs().sum()) > 0
