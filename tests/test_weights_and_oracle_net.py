"""CPU checks of the host logic around the network path:
  * the architecture tables reproduce SURVEY.md 8(a)'s counts (94 trunk convs, 12.393 GFLOP per crop, 100.2 M params);
  * BatchNorm folding + (cout, K) packing equals the unfused reference formulation;
  * every TF operator restated in oracle/net.py agrees with an independent naive numpy implementation.
"""
import numpy as np
import pytest
import torch

from monopsr_amd.core import weights as W
from oracle import net as onet


def test_architecture_counts_match_survey():
    specs = W.trunk_conv_specs()
    assert len(specs) == 94
    macs = 0
    for s in specs:
        pos = 24 * 24 if s["role"] == "root" else 144
        macs += pos * s["kh"] * s["kw"] * s["cin"] * s["cout"]
    assert macs == 3957000000 + 0 or abs(macs - 3.957e9) < 2e6  # 3.957 GMAC per crop
    dec = 0
    for name, kh, kw, cin, cout, _, _, _ in W.DECODER_SPECS:
        pos = 144 if name.startswith("squash") else (576 if "conv2" in name else 2304)
        dec += pos * kh * kw * cin * cout
    heads = sum(fin * fout for _, fin, fout, _ in W.head_fc_specs())
    assert macs + dec + heads == 6196658176  # 12.393 GFLOP = 2 x this


def test_parameter_count_matches_survey():
    w = W.synthetic_weights(seed=0, scopes=(W.CROP_SCOPE, W.FULL_SCOPE))
    n = sum(v.size for k, v in w.items() if "BatchNorm" not in k)
    assert n == 100204832  # conv/FC weights + biases: the gradient all-reduce size quoted in SURVEY 5 / 8(e)
    bn = sum(v.size for k, v in w.items() if "BatchNorm" in k and "moving_" not in k)
    assert bn == 2 * 2 * 41408 + 768  # trunk gamma+beta (2 trunks) + decoder beta: trainable too, 0.17 % more


def test_layer_tables_have_the_abi_lengths():
    w = W.synthetic_weights(seed=1, width_div=2)
    assert len(W.pack_trunk(w, W.CROP_SCOPE, 2)[1]) == 94
    assert len(W.pack_decoder(w, 2)[1]) == 7
    blob, recs = W.pack_heads(w, 36 * 256)
    assert len(recs) == 7
    assert [r["cout"] for r in recs] == [2048, 1024, 1024, 27, 1024, 1024, 2]
    assert recs[1]["cin"] == 1056 and recs[4]["cin"] == 1088  # 1043 / 1060 zero-padded to a multiple of 32
    for r in recs:
        assert r["w_off"] % 64 == 0 and (r["b_off"] % 64 == 0)


@pytest.mark.parametrize("with_gamma", [True, False])
def test_bn_fold_equals_unfused(with_gamma):
    rng = np.random.default_rng(3)
    x = torch.from_numpy(rng.standard_normal((2, 6, 6, 8)).astype(np.float32))
    w = rng.standard_normal((3, 3, 8, 5)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, 5).astype(np.float32) if with_gamma else None
    beta, mean = rng.normal(0, 0.1, 5).astype(np.float32), rng.normal(0, 0.1, 5).astype(np.float32)
    var = rng.uniform(0.5, 1.5, 5).astype(np.float32)
    ref = onet.tf_batch_norm(onet.tf_conv2d(x, torch.from_numpy(w), rate=2),
                             torch.from_numpy(gamma) if with_gamma else None, torch.from_numpy(beta),
                             torch.from_numpy(mean), torch.from_numpy(var), 1e-3)
    w_ok, b = W.fold_conv(w, gamma, beta, mean, var, 1e-3)
    w_back = torch.from_numpy(w_ok.reshape(5, 3, 3, 8).transpose(1, 2, 3, 0).copy())
    got = onet.tf_conv2d(x, w_back, rate=2) + torch.from_numpy(b)
    np.testing.assert_allclose(got.numpy(), ref.numpy(), rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------ TF operator restatements vs naive numpy

def _naive_conv_same(x, w, rate):
    b, h, wd, c = x.shape
    kh, kw, _, n = w.shape
    out = np.zeros((b, h, wd, n), np.float64)
    for y in range(h):
        for xx in range(wd):
            for ky in range(kh):
                for kx in range(kw):
                    sy, sx = y + (ky - kh // 2) * rate, xx + (kx - kw // 2) * rate
                    if 0 <= sy < h and 0 <= sx < wd:
                        out[:, y, xx] += x[:, sy, sx].astype(np.float64) @ w[ky, kx].astype(np.float64)
    return out


@pytest.mark.parametrize("rate", [1, 2, 4])
def test_conv_same_atrous_vs_naive(rate):
    rng = np.random.default_rng(rate)
    x = rng.standard_normal((2, 7, 9, 3)).astype(np.float32)
    w = rng.standard_normal((3, 3, 3, 4)).astype(np.float32)
    got = onet.tf_conv2d(torch.from_numpy(x), torch.from_numpy(w), rate=rate).numpy()
    np.testing.assert_allclose(got, _naive_conv_same(x, w, rate), rtol=1e-4, atol=1e-5)


def test_root_conv_is_pad3_stride2_valid():
    """resnet_utils.conv2d_same with stride 2: explicit pad (3,3) then VALID -> 48 -> 24; differs from SAME."""
    rng = np.random.default_rng(5)
    x = rng.standard_normal((1, 48, 48, 3)).astype(np.float32)
    w = rng.standard_normal((7, 7, 3, 2)).astype(np.float32)
    xp = np.pad(x, ((0, 0), (3, 3), (3, 3), (0, 0)))
    ref = np.zeros((1, 24, 24, 2))
    for y in range(24):
        for xx in range(24):
            ref[0, y, xx] = np.tensordot(xp[0, 2 * y:2 * y + 7, 2 * xx:2 * xx + 7].astype(np.float64),
                                         w.astype(np.float64), axes=3)
    xt = torch.nn.functional.pad(torch.from_numpy(x).permute(0, 3, 1, 2), (3, 3, 3, 3)).permute(0, 2, 3, 1)
    got = onet.tf_conv2d(xt, torch.from_numpy(w), stride=2, padding="VALID").numpy()
    assert got.shape == (1, 24, 24, 2)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


def test_max_pool_same_pads_bottom_right():
    x = np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1)
    got = onet.tf_max_pool(torch.from_numpy(x), 3, 2, "SAME").numpy()[0, :, :, 0]
    # out 2x2, total pad 1 -> nothing on top/left, one row/col at bottom/right
    np.testing.assert_array_equal(got, [[10, 11], [14, 15]])
    x = -np.arange(1, 10, dtype=np.float32).reshape(1, 3, 3, 1)  # negatives: padding must not win
    got = onet.tf_max_pool(torch.from_numpy(x), 3, 2, "SAME").numpy()[0, :, :, 0]
    np.testing.assert_array_equal(got, [[-1, -2], [-4, -5]])
    got = onet.tf_max_pool(torch.from_numpy(np.arange(16, dtype=np.float32).reshape(1, 4, 4, 1)), 2, 2, "VALID")
    np.testing.assert_array_equal(got.numpy()[0, :, :, 0], [[5, 7], [13, 15]])


def _naive_resize(x, oh, ow, ac):
    b, h, w, c = x.shape
    out = np.zeros((b, oh, ow, c), np.float64)
    hs = (h - 1) / (oh - 1) if (ac and oh > 1) else h / oh
    ws = (w - 1) / (ow - 1) if (ac and ow > 1) else w / ow
    for y in range(oh):
        sy = np.float32(y) * np.float32(hs)
        y0 = int(np.floor(sy)); y1 = min(y0 + 1, h - 1); yl = float(sy - y0)
        for xx in range(ow):
            sx = np.float32(xx) * np.float32(ws)
            x0 = int(np.floor(sx)); x1 = min(x0 + 1, w - 1); xl = float(sx - x0)
            top = x[:, y0, x0] + (x[:, y0, x1] - x[:, y0, x0]) * xl
            bot = x[:, y1, x0] + (x[:, y1, x1] - x[:, y1, x0]) * xl
            out[:, y, xx] = top + (bot - top) * yl
    return out


@pytest.mark.parametrize("shape,out,ac", [((1, 12, 12, 2), (24, 24), True), ((1, 9, 31, 3), (8, 30), False),
                                          ((2, 5, 6, 1), (11, 4), True)])
def test_resize_bilinear_vs_naive(shape, out, ac):
    x = np.random.default_rng(7).standard_normal(shape).astype(np.float32)
    got = onet.tf_resize_bilinear(torch.from_numpy(x), out[0], out[1], ac).numpy()
    np.testing.assert_allclose(got, _naive_resize(x, out[0], out[1], ac), rtol=1e-5, atol=1e-5)


def test_resize_align_corners_hits_the_corners():
    x = np.random.default_rng(8).standard_normal((1, 12, 12, 4)).astype(np.float32)
    got = onet.tf_resize_bilinear(torch.from_numpy(x), 24, 24, True).numpy()
    np.testing.assert_allclose(got[0, 0, 0], x[0, 0, 0], rtol=1e-6)
    np.testing.assert_allclose(got[0, 23, 23], x[0, 11, 11], rtol=1e-5)


def _naive_crop(image, box, ch, cw, extrap=0.0):
    h, w, c = image.shape
    y1, x1, y2, x2 = [np.float32(v) for v in box]
    out = np.full((ch, cw, c), extrap, np.float64)
    for i in range(ch):
        in_y = (y1 * np.float32(h - 1) + np.float32(i) * ((y2 - y1) * np.float32(h - 1) / np.float32(ch - 1))
                if ch > 1 else np.float32(0.5) * (y1 + y2) * np.float32(h - 1))
        if in_y < 0 or in_y > h - 1:
            continue
        t, bt, yl = int(np.floor(in_y)), int(np.ceil(in_y)), float(in_y - np.floor(in_y))
        for j in range(cw):
            in_x = (x1 * np.float32(w - 1) + np.float32(j) * ((x2 - x1) * np.float32(w - 1) / np.float32(cw - 1))
                    if cw > 1 else np.float32(0.5) * (x1 + x2) * np.float32(w - 1))
            if in_x < 0 or in_x > w - 1:
                continue
            l, r, xl = int(np.floor(in_x)), int(np.ceil(in_x)), float(in_x - np.floor(in_x))
            top = image[t, l] + (image[t, r] - image[t, l]) * xl
            bot = image[bt, l] + (image[bt, r] - image[bt, l]) * xl
            out[i, j] = top + (bot - top) * yl
    return out


def test_crop_and_resize_vs_naive():
    rng = np.random.default_rng(9)
    img = rng.standard_normal((2, 13, 17, 3)).astype(np.float32)
    boxes = np.array([[0.1, 0.2, 0.7, 0.9], [0, 0, 1, 1], [-0.3, 0.2, 0.5, 1.4], [0.8, 0.9, 0.1, 0.2]], np.float32)
    ind = np.array([0, 1, 1, 0])
    got = onet.tf_crop_and_resize(torch.from_numpy(img), boxes, ind, 6, 5).numpy()
    for k in range(4):
        np.testing.assert_allclose(got[k], _naive_crop(img[ind[k]], boxes[k], 6, 5), rtol=1e-5, atol=1e-6)
    # identity box reproduces the image when the crop has the image's size
    same = onet.tf_crop_and_resize(torch.from_numpy(img), np.array([[0, 0, 1, 1]], np.float32), [0], 13, 17).numpy()
    np.testing.assert_allclose(same[0], img[0], rtol=1e-5, atol=1e-6)


def test_output_stride_4_structure():
    """Narrow copy of the trunk: 48x48 crop -> 12x12 block3 map; block3 uses dilation 4 (a unit impulse placed
    5 pixels from the probe changes it only through the stacked atrous 3x3s, never through stride)."""
    w = W.synthetic_weights(seed=2, width_div=8, decoder=False, heads=False)
    x = torch.zeros((1, 48, 48, 3))
    c = {}
    y = onet.resnet101_block3(x, w, W.CROP_SCOPE, c)
    assert tuple(c["pool1"].shape) == (1, 12, 12, 8)
    assert tuple(c["block1"].shape) == (1, 12, 12, 32)
    assert tuple(c["block2"].shape) == (1, 12, 12, 64)
    assert tuple(y.shape) == (1, 12, 12, 128)


def test_heads_feature_concat_sizes_and_one_hot():
    """1043 / 1060 inputs; class index 1 with one class -> all-zero one-hot (tf.one_hot)."""
    w = W.synthetic_weights(seed=4, trunk=False, decoder=False)
    assert w["output/proposal_fc/proposal_fc/fc0/weights"].shape == (1043, 1024)
    assert w["output/regression_fc/regression_fc/fc0/weights"].shape == (1060, 1024)
    B = 2
    feat = torch.relu(torch.randn(B, 6, 6, 512))
    boxes = np.array([[100, 200, 180, 330], [50, 700, 210, 900]], np.float32)
    cam_p = np.array([[721.5, 0, 609.5, 44.8], [0, 721.5, 172.8, 0.2], [0, 0, 1, 0.003]], np.float32)
    args = (boxes, cam_p, np.array([0.1, -0.2], np.float32))
    o1 = onet.heads(feat, *args, np.array([[1], [1]]), np.ones((B, 3), np.float32), np.full(B, 2.178, np.float32), w)
    o0 = onet.heads(feat, *args, np.array([[0], [0]]), np.ones((B, 3), np.float32), np.full(B, 2.178, np.float32), w)
    assert not torch.allclose(o1["lwh"], o0["lwh"])  # index 0 switches the one-hot on, index 1 leaves it off
    z = 721.5 * o1["lwh"][:, 2] / torch.tensor([80.0, 160.0]) + 2.178
    torch.testing.assert_close(o1["prop_cen_z"][:, 0], z, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(o1["centroids"][:, 0:1], o1["cen_z"] * torch.tan(torch.tensor([[0.1], [-0.2]])) -
                               44.8 / 721.5, rtol=1e-5, atol=1e-5)
