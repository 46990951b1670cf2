"""BASELINE configs 2 and 3 at their full per-GPU size (256 instances), the full-width full-image trunk, and a batch
that really exceeds DeviceNet.MAX_CHUNK -- through the C ABI, against the oracle on slices the oracle finishes in
seconds and through size-independent properties (instances are independent; results are deterministic).

Why chunk-vs-batch comparisons use a tolerance instead of bit identity: for small launches the library cuts the K loop
into slices (split-K / stream-K) to fill the chip, so the fp32 summation ORDER of an output element depends on the
batch size; every order is a valid evaluation of the same sum (differences ~1e-6 of the tensor scale).
"""
import numpy as np
import pytest
import torch

import bench
from oracle import net as onet
from oracle import ops as orc

pytestmark = pytest.mark.gpu

B, NPTS = 256, 1024


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu() if isinstance(b, torch.Tensor) else torch.as_tensor(b).double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


@pytest.fixture(scope="module")
def full_batch():
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    dev = torch.device("cuda", 0)
    weights = W.synthetic_weights(seed=0)
    net = dn.DeviceNet(weights, device=dev)
    inp, host = bench.make_inputs(B, NPTS, 0, dev)
    step = bench.Step(net, inp, NPTS)
    xyz, out = step.forward_net()
    res = {k: v.clone() for k, v in out.items()}
    res["xyz"] = xyz.clone()
    torch.cuda.synchronize()
    return weights, net, inp, host, step, res


def test_cfg2_network_leg_against_the_oracle_on_a_slice(full_batch):
    """ResNet-101 trunk + squash/decoder/xyz + heads at batch 256 (BASELINE config 2): instances 0, 100, 101 and 255
    of the batch against oracle/net.py run on exactly those instances."""
    weights, net, inp, host, step, res = full_batch
    pick = np.array([0, 100, 101, 255])
    ref = onet.instance_path(host["crops"][pick], host["full_feat"][pick], host["boxes"][pick], bench.P2,
                             host["view"][pick], np.ones((len(pick), 1), np.int32),
                             inp["mean_lwh"][pick].cpu().numpy(), inp["z_off"][pick].cpu().numpy(), weights)
    idx = torch.as_tensor(pick, device="cuda")
    assert _rel(res["xyz"][idx], ref["inst_xyz_map_local"]) < 1e-4
    assert _rel(res["centroids"][idx], ref["centroids"]) < 1e-4
    assert _rel(res["lwh"][idx], ref["lwh"]) < 1e-4
    assert _rel(res["alpha_bins"][idx], ref["alpha_bins"]) < 1e-4


def test_cfg2_instances_do_not_depend_on_their_batch(full_batch):
    """Instances 96..103 run alone (a batch of 8) give what they give inside the batch of 256, and the batch of 256
    gives the same bits when run again."""
    weights, net, inp, host, step, res = full_batch
    lo, hi = 96, 104
    small = {k: (v[lo:hi].contiguous() if (v.dim() > 0 and v.shape[0] == B and k != "cam_p") else v)
             for k, v in inp.items()}
    xyz8, out8 = bench.Step(net, small, NPTS).forward_net()
    assert _rel(xyz8, res["xyz"][lo:hi]) < 2e-5
    for k in ("centroids", "lwh", "alpha_bins", "alpha_regs", "cen_z"):
        assert _rel(out8[k], res[k][lo:hi]) < 2e-5, k
    xyz_again, out_again = step.forward_net()
    assert torch.equal(xyz_again, res["xyz"]) and torch.equal(out_again["centroids"], res["centroids"])


def test_cfg3_chamfer_leg_full_batch(full_batch):
    """+ 1024-point nn_distance forward / backward on the batch's own predicted clouds (BASELINE config 3): distances
    and indices bit-exact against the C oracle on a 6-cloud slice, gradients to 1e-5, and the op's properties on all
    256 clouds."""
    from monopsr_amd.tf_ops.nn_distance import tf_nndistance as nnd
    weights, net, inp, host, step, res = full_batch
    pred = res["xyz"].reshape(B, -1, 3)[:, :NPTS].contiguous()
    d1, i1, d2, i2 = nnd.nn_distance(pred, inp["gt"])
    ones = torch.ones_like(d1)
    g1, g2 = nnd.nn_distance_grad(pred, inp["gt"], ones, i1, ones, i2)
    s = slice(120, 126)
    p, g = pred[s].cpu().numpy(), host["gt"][s]
    rd1, ri1, rd2, ri2 = orc.nn_distance(p, g)
    assert (d1[s].cpu().numpy() == rd1).all() and (i1[s].cpu().numpy() == ri1).all()
    assert (d2[s].cpu().numpy() == rd2).all() and (i2[s].cpu().numpy() == ri2).all()
    r1, r2 = orc.nn_distance_grad(p, g, np.ones_like(rd1), ri1, np.ones_like(rd2), ri2)
    np.testing.assert_allclose(g1[s].cpu().numpy(), r1, rtol=0, atol=1e-5 * np.abs(r1).max())
    np.testing.assert_allclose(g2[s].cpu().numpy(), r2, rtol=0, atol=1e-5 * np.abs(r2).max())
    # properties at full size: the reported neighbour really is at the reported distance; no other point is closer
    # (checked for a random subset of queries of every cloud)
    q = torch.randint(0, NPTS, (B, 16), device="cuda")
    pq = torch.gather(pred, 1, q[:, :, None].expand(-1, -1, 3))
    nb = torch.gather(inp["gt"], 1, torch.gather(i1, 1, q).long()[:, :, None].expand(-1, -1, 3))
    dd = ((nb - pq) ** 2).sum(-1)
    torch.testing.assert_close(dd, torch.gather(d1, 1, q), rtol=1e-5, atol=1e-6)
    allq = ((inp["gt"][:, None, :, :] - pq[:, :, None, :]) ** 2).sum(-1)  # (B,16,NPTS)
    assert bool((allq.min(-1).values >= torch.gather(d1, 1, q) * (1 - 1e-5) - 1e-6).all())
    # the gradient of sum(dist1) + sum(dist2) w.r.t. a joint translation of ONE cloud pair is zero-sum
    torch.testing.assert_close(g1.sum(1) + g2.sum(1), torch.zeros((B, 3), device="cuda"), rtol=0,
                               atol=1e-3 * float(g1.abs().sum(1).max()))


def test_full_image_trunk_full_width_vs_oracle():
    """SURVEY 8(a) a3 at full width: the second ResNet-101 on a (1,160,608,3) image -> (1,40,152,1024)."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    weights = W.synthetic_weights(seed=5, decoder=False, heads=False, scopes=(W.FULL_SCOPE,))
    img = (np.random.default_rng(5).standard_normal((1, 160, 608, 3)) * 50).astype(np.float32)
    with torch.no_grad():
        ref = onet.resnet101_block3(torch.from_numpy(img), weights, W.FULL_SCOPE)
    net = dn.DeviceNet.__new__(dn.DeviceNet)
    net.device = torch.device("cuda")
    net.crop_trunk = dn.PackedPart(*W.pack_trunk(weights, W.FULL_SCOPE, 1), net.device)
    net.full_trunk = None
    net.ws_trunk = dn.Workspace(net.device)
    got = net.trunk(torch.from_numpy(img).cuda())
    assert tuple(got.shape) == (1, 40, 152, 1024)
    assert _rel(got, ref) < 1e-4


def test_a_batch_above_max_chunk():
    """520 instances > DeviceNet.MAX_CHUNK = 512: two native calls (512 + 8); the first and the last eight instances
    equal what they give on their own."""
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    assert dn.DeviceNet.MAX_CHUNK == 512
    weights = W.synthetic_weights(seed=0)
    net = dn.DeviceNet(weights)
    n = 520
    g = torch.Generator(device="cuda").manual_seed(9)
    crops = torch.randn((n, 48, 48, 3), device="cuda", generator=g) * 50
    full = torch.relu(torch.randn((n, 12, 12, 1024), device="cuda", generator=g))
    feat = net.trunk(crops)
    fb, _, xyz = net.squash_decoder(feat, full, (48, 48), want_feat_map=False)
    assert feat.shape[0] == n and xyz.shape[0] == n
    for lo in (0, 512):
        f8 = net.trunk(crops[lo:lo + 8].contiguous())
        assert _rel(f8, feat[lo:lo + 8]) < 2e-5
        _, _, x8 = net.squash_decoder(f8, full[lo:lo + 8].contiguous(), (48, 48), want_feat_map=False)
        assert _rel(x8, xyz[lo:lo + 8]) < 2e-5


@pytest.mark.parametrize("want_feat_map", [False, True])
def test_decoder_channel_blocked_layout_is_bit_identical(want_feat_map):
    """mpsr_squash_decoder_fwd keeps its internal tensors channel-blocked ([C/8][H][W][8]) when all four 3x3 layers run
    on the F(4x4,3x3) kernel (network.hip; the batch must be large enough for the library to pick that kernel): same
    kernels and the same order of arithmetic as the NHWC chain, so every output must agree bit for bit -- with the
    feature map requested (last conv writes NHWC, xyz head reads NHWC) and without (both channel-blocked)."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    lib = _lib.lib()
    B = 128
    net = dn.DeviceNet(W.synthetic_weights(seed=3, width_div=2), width_div=2)
    g = torch.Generator(device="cuda").manual_seed(11)
    crop = torch.randn((B, 12, 12, 512), device="cuda", generator=g).clamp_(min=0)
    full = torch.randn((B, 12, 12, 512), device="cuda", generator=g).clamp_(min=0)
    outs = {}
    lib.mpsr_debug_set_decoder_upconv(0)  # (like with like: the upsampled layers on resize + F(4x4,3x3) in both runs)
    try:
        for on in (1, 0):
            lib.mpsr_debug_set_decoder_c8(on)
            try:
                outs[on] = [t.clone() if t is not None else None
                            for t in net.squash_decoder(crop, full, (48, 48), want_feat_map=want_feat_map)]
            finally:
                lib.mpsr_debug_set_decoder_c8(1)
    finally:
        lib.mpsr_debug_set_decoder_upconv(1)
    for a, b in zip(outs[1], outs[0]):
        assert (a is None) == (b is None)
        if a is not None:
            assert torch.equal(a, b)
    assert float(outs[1][2].abs().max()) > 0


@pytest.mark.parametrize("width_div,B", [(1, 64), (2, 128)])
def test_decoder_upsampled_convs_as_tap_gemm_vs_resize_winograd(width_div, B):
    """conv2_1 / conv3_1 read a bilinearly upsampled map (net_builder.py:72-77, :81-85).  Default: tap GEMM on the source
    map + gather (csrc/upconv.hip), no upsampled tensor; mpsr_debug_set_decoder_upconv(0): resize + F(4x4,3x3) as in
    round 3.  Same operator, different arithmetic order: every output within 1e-4 of the tensor scale (measured ~2e-5,
    the F(4x4) side's error), deterministic, and the filter cache (second call) changes nothing.  width_div 2: only
    conv2_1 qualifies (conv3_1 has 64 outputs), the chain mixes both forms."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    lib = _lib.lib()
    net = dn.DeviceNet(W.synthetic_weights(seed=5, width_div=width_div), width_div=width_div)
    g = torch.Generator(device="cuda").manual_seed(12)
    c = 1024 // width_div
    crop = torch.randn((B, 12, 12, c), device="cuda", generator=g).clamp_(min=0)
    full = torch.randn((B, 12, 12, c), device="cuda", generator=g).clamp_(min=0)
    outs = {}
    for on in (1, 0):
        lib.mpsr_debug_set_decoder_upconv(on)
        try:
            net.fcache = {}
            first = [t.clone() for t in net.squash_decoder(crop, full, (48, 48), want_feat_map=True)]
            outs[on] = [t.clone() for t in net.squash_decoder(crop, full, (48, 48), want_feat_map=True)]  # cache valid
            for a, b in zip(first, outs[on]):
                assert torch.equal(a, b)
        finally:
            lib.mpsr_debug_set_decoder_upconv(1)
    for name, a, b in zip(("features_for_box_3d", "features_for_map", "inst_xyz_map_local"), outs[1], outs[0]):
        assert _rel(a, b) < 1e-4, (name, _rel(a, b))
    assert float(outs[1][2].abs().max()) > 0


def test_filter_cache_survives_changes_of_mode_and_batch():
    """DeviceNet keeps the transformed filters of its 3x3 layers across calls (mpsr_net_opts.filter_cache).  WHICH form a
    layer's slice holds depends on the kernel a call picks -- F(4x4,3x3) filters, the tap GEMM's re-ordered rows, nothing
    -- and that changes with the arithmetic mode, the Winograd policy and the batch size; the library notes the form
    per layer (filter_cache_tags) and re-fills a slice that holds another one.  One net, calls in every order: each
    configuration must reproduce its own first result bit for bit, and the fp32 ones must agree with each other."""
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    net = dn.DeviceNet(W.synthetic_weights(seed=13))
    g = torch.Generator(device="cuda").manual_seed(14)
    feats = {B: (torch.randn((B, 12, 12, 1024), device="cuda", generator=g).clamp_(min=0),
                 torch.randn((B, 12, 12, 1024), device="cuda", generator=g).clamp_(min=0)) for B in (128, 8)}

    def run(B, math, policy):
        _lib.set_conv_math(math)
        _lib.set_winograd_policy(policy)
        try:
            return [t.clone() for t in net.squash_decoder(feats[B][0], feats[B][1], (48, 48), want_feat_map=True)]
        finally:
            _lib.set_conv_math("fp32")
            _lib.set_winograd_policy("auto")
    configs = [(128, "fp32", "auto"), (128, "bf16x3", "auto"), (128, "fp32", "off"), (8, "fp32", "auto"), (8, "bf16x3", "auto")]
    first = {}
    for order in (configs, configs[::-1], configs[1:] + configs[:1]):
        for cfg in order:
            out = run(*cfg)
            if cfg not in first:
                first[cfg] = out
            for a, b in zip(out, first[cfg]):
                assert torch.equal(a, b), cfg
    for a, b in zip(first[(128, "fp32", "auto")], first[(128, "fp32", "off")]):
        assert _rel(a, b) < 1e-4
    for a, b in zip(first[(128, "fp32", "auto")], first[(128, "bf16x3", "auto")]):
        assert _rel(a, b) < 1e-3
