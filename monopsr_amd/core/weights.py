"""Architecture tables, synthetic weights and weight packing for the MonoPSR instance path.

Variables carry the reference's TF names and layouts (conv kernels HWIO, FC (in, out), BatchNorm
gamma/beta/moving_mean/moving_variance) so that a TF-checkpoint importer only has to fill the same dict
(scopes: checkpoint_utils.py:82-106, net_builder.py:44-50,63-89, monopsr_output_builder.py:95-104,126-302,457-661).

`pack_*` fold inference-mode BatchNorm into the convolution and lay weights out as the HIP kernels read them:
(cout, kh*kw*cin) row-major, K contiguous, one flat fp32 blob + a table of mpsr_layer records
(include/monopsr_hip.h).  numpy only -- no torch, no oracle.
"""
import numpy as np

CROP_SCOPE = "FirstStageFeatureExtractor_crop/resnet_v1_101"
FULL_SCOPE = "FirstStageFeatureExtractor_full/resnet_v1_101"

# resnet_v1_101 blocks executed up to the returned handle block3 (resnet_v1.py:320-324,
# faster_rcnn_resnet_v1_feature_extractor.py:244-245): (name, bottleneck depth, units, atrous rate at output_stride 4)
TRUNK_BLOCKS = (("block1", 64, 3, 1), ("block2", 128, 4, 2), ("block3", 256, 23, 4))
TRUNK_BN_EPS = 1e-5    # faster_rcnn_resnet_v1_feature_extractor.py:229-232
DECODER_BN_EPS = 1e-3  # slim.batch_norm default, net_builder.py:77-79
ROOT_K = 147           # 7*7*3
ROOT_KPAD = 160        # im2col columns, padded to a multiple of 32 for the MFMA K loop


def trunk_conv_specs(scope=CROP_SCOPE):
    """Convolutions of the trunk in execution order: dicts(name, kh, kw, cin, cout, rate, relu, role)."""
    specs = [dict(name=scope + "/conv1", kh=7, kw=7, cin=3, cout=64, rate=1, relu=True, role="root")]
    cin = 64
    for block, depth_b, units, rate in TRUNK_BLOCKS:
        depth = depth_b * 4
        for u in range(1, units + 1):
            p = "%s/%s/unit_%d/bottleneck_v1" % (scope, block, u)
            if cin != depth:  # projection shortcut, BN, no activation (resnet_v1.py:110-119)
                specs.append(dict(name=p + "/shortcut", kh=1, kw=1, cin=cin, cout=depth, rate=1, relu=False,
                                  role="shortcut"))
            specs.append(dict(name=p + "/conv1", kh=1, kw=1, cin=cin, cout=depth_b, rate=1, relu=True, role="conv1"))
            specs.append(dict(name=p + "/conv2", kh=3, kw=3, cin=depth_b, cout=depth_b, rate=rate, relu=True,
                              role="conv2"))
            # conv3 has no activation of its own; ReLU follows the residual add (resnet_v1.py:125-133)
            specs.append(dict(name=p + "/conv3", kh=1, kw=1, cin=depth_b, cout=depth, rate=1, relu=True,
                              role="conv3"))
            cin = depth
    return specs


DECODER_SPECS = (
    # name, kh, kw, cin, cout, has_bn, has_bias, relu
    ("squash/1x1_conv", 1, 1, 2048, 512, False, True, True),
    ("map_decoder/conv2/conv2_1", 3, 3, 512, 256, True, False, True),
    ("map_decoder/conv2/conv2_2", 3, 3, 256, 256, True, False, True),
    ("map_decoder/conv3/conv3_1", 3, 3, 256, 128, True, False, True),
    ("map_decoder/conv3/conv3_2", 3, 3, 128, 128, True, False, True),
    ("output/inst_xyz_map_local/inst_xyz_map_local", 3, 3, 128, 3, False, True, False),
)


def head_fc_specs(feat_elems=18432, num_classes=1, num_alpha_bins=12, fc_sizes=(1024, 1024)):
    """Fully-connected layers of the heads: (name, in, out, relu)."""
    prop_in = 1024 + 4 + 1 + 1 + num_classes + 12
    reg_in = 1024 + 4 + 1 + 1 + num_classes + 3 + 2 * num_alpha_bins + 2
    specs = [("output/proposal_fc/proposal_fc/img_fc", feat_elems, 1024, True)]
    fin = prop_in
    for i, size in enumerate(fc_sizes):
        specs.append(("output/proposal_fc/proposal_fc/fc%d" % i, fin, size, True))
        fin = size
    specs.append(("output/lwh/lwh", fin, 3, False))
    specs.append(("output/alpha", fin, 2 * num_alpha_bins, False))
    specs.append(("output/regression_fc/regression_fc/img_fc", feat_elems, 1024, True))
    fin = reg_in
    for i, size in enumerate(fc_sizes):
        specs.append(("output/regression_fc/regression_fc/fc%d" % i, fin, size, True))
        fin = size
    specs.append(("output/cen_y/cen_y", fin, 1, False))
    specs.append(("output/cen_z_offs/cen_z", fin, 1, False))
    return specs


# ------------------------------------------------------------------------------------------- synthetic weights

def _bn(rng, c, with_gamma):
    d = {"beta": rng.normal(0, 0.1, c).astype(np.float32),
         "moving_mean": rng.normal(0, 0.1, c).astype(np.float32),
         "moving_variance": rng.uniform(0.5, 1.5, c).astype(np.float32)}
    if with_gamma:
        d["gamma"] = rng.uniform(0.5, 1.5, c).astype(np.float32)
    return d


def synthetic_weights(seed=0, width_div=1, scopes=(CROP_SCOPE,), feat_elems=None, trunk=True, decoder=True,
                      heads=True):
    """Seeded random-init weights of the reference architecture: variance-scaling convolutions,
    BN gamma ~ U(0.5,1.5), beta/mean ~ N(0,0.1), variance ~ U(0.5,1.5), Xavier-uniform FC, biases ~ N(0,0.1).
    width_div > 1 divides every channel count (tests use narrow copies of the same graph)."""
    rng = np.random.default_rng(seed)
    w = {}
    if trunk:
        for scope in scopes:
            for s in scaled_trunk_specs(scope, width_div):
                fan_in = s["kh"] * s["kw"] * s["cin"]
                w[s["name"] + "/weights"] = (rng.standard_normal((s["kh"], s["kw"], s["cin"], s["cout"])) *
                                             np.sqrt(2.0 / fan_in)).astype(np.float32)
                for k, v in _bn(rng, s["cout"], True).items():
                    if k == "gamma" and s["role"] == "conv3":
                        v = v * np.float32(0.2)  # damp the residual branch so 30 stacked units keep O(10^2) activations
                    w[s["name"] + "/BatchNorm/" + k] = v
    if decoder:
        for name, kh, kw, cin, cout, has_bn, has_bias, _ in scaled_decoder_specs(width_div):
            fan_in, fan_out = kh * kw * cin, kh * kw * cout
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            w[name + "/weights"] = rng.uniform(-lim, lim, (kh, kw, cin, cout)).astype(np.float32)
            if has_bn:
                for k, v in _bn(rng, cout, False).items():
                    w[name + "/BatchNorm/" + k] = v
            if has_bias:
                w[name + "/biases"] = rng.normal(0, 0.1, cout).astype(np.float32)
    if heads:
        if feat_elems is None:
            feat_elems = 36 * (512 // width_div)
        for name, fin, fout, _ in head_fc_specs(feat_elems):
            lim = np.sqrt(6.0 / (fin + fout))
            w[name + "/weights"] = rng.uniform(-lim, lim, (fin, fout)).astype(np.float32)
            w[name + "/biases"] = rng.normal(0, 0.1, fout).astype(np.float32)
    return w


def scaled_trunk_specs(scope=CROP_SCOPE, width_div=1):
    specs = trunk_conv_specs(scope)
    if width_div == 1:
        return specs
    out = []
    for s in specs:
        s = dict(s)
        if s["role"] != "root":
            s["cin"] //= width_div
        s["cout"] //= width_div
        out.append(s)
    return out


def scaled_decoder_specs(width_div=1):
    out = []
    for name, kh, kw, cin, cout, has_bn, has_bias, relu in DECODER_SPECS:
        out.append((name, kh, kw, cin // width_div, cout if cout == 3 else cout // width_div, has_bn, has_bias, relu))
    return out


# ------------------------------------------------------------------------------------------- packing

class Blob:
    """Flat fp32 weight blob + layer table under construction."""

    def __init__(self):
        self.chunks = []
        self.size = 0
        self.layers = []  # dicts with the mpsr_layer fields

    def _push(self, arr):
        arr = np.ascontiguousarray(arr, np.float32).ravel()
        off = self.size
        pad = (-arr.size) % 64  # keep every tensor 256-byte aligned
        self.chunks.append(arr)
        if pad:
            self.chunks.append(np.zeros(pad, np.float32))
        self.size += arr.size + pad
        return off

    def add_layer(self, w_ok, bias, cin, kh, kw, dilation, relu):
        """w_ok: (cout, kh*kw*cin)."""
        cout = w_ok.shape[0]
        assert w_ok.shape[1] == kh * kw * cin, (w_ok.shape, kh, kw, cin)
        rec = dict(cin=cin, cout=cout, kh=kh, kw=kw, dilation=dilation, relu=int(relu), w_off=self._push(w_ok),
                   b_off=self._push(bias) if bias is not None else -1)
        self.layers.append(rec)
        return rec

    def finish(self):
        data = np.concatenate(self.chunks) if self.chunks else np.zeros(0, np.float32)
        return data, list(self.layers)


def fold_conv(w_hwio, gamma=None, beta=None, mean=None, var=None, eps=0.0, bias=None):
    """HWIO kernel (+ inference BatchNorm or bias) -> ((cout, kh*kw*cin) fp32, (cout) fp32 bias).
    BN: y = gamma*(x-mean)/sqrt(var+eps)+beta  ==  conv(x, w*s) + (beta - mean*s), s = gamma/sqrt(var+eps);
    the fold is done in float64 and rounded once."""
    kh, kw, cin, cout = w_hwio.shape
    w = w_hwio.astype(np.float64).transpose(3, 0, 1, 2).reshape(cout, kh * kw * cin)
    if var is not None:
        s = 1.0 / np.sqrt(var.astype(np.float64) + eps)
        if gamma is not None:
            s = s * gamma.astype(np.float64)
        b = beta.astype(np.float64) - mean.astype(np.float64) * s
        w = w * s[:, None]
    else:
        b = bias.astype(np.float64) if bias is not None else None
    return w.astype(np.float32), (b.astype(np.float32) if b is not None else None)


def _pad_k(w_ok, kpad):
    out = np.zeros((w_ok.shape[0], kpad), np.float32)
    out[:, :w_ok.shape[1]] = w_ok
    return out


def pack_trunk(weights, scope=CROP_SCOPE, width_div=1):
    """-> (blob fp32, layer records) in the order mpsr_trunk_fwd consumes them."""
    blob = Blob()
    for s in scaled_trunk_specs(scope, width_div):
        n = s["name"]
        w_ok, b = fold_conv(weights[n + "/weights"], weights[n + "/BatchNorm/gamma"], weights[n + "/BatchNorm/beta"],
                            weights[n + "/BatchNorm/moving_mean"], weights[n + "/BatchNorm/moving_variance"],
                            TRUNK_BN_EPS)
        if s["role"] == "root":  # consumed as a 1x1 layer over the im2col matrix
            blob.add_layer(_pad_k(w_ok, ROOT_KPAD), b, ROOT_KPAD, 1, 1, 1, True)
        else:
            blob.add_layer(w_ok, b, s["cin"], s["kh"], s["kw"], s["rate"], s["relu"])
    return blob.finish()


def pack_decoder(weights, width_div=1):
    blob = Blob()
    for name, kh, kw, cin, cout, has_bn, has_bias, relu in scaled_decoder_specs(width_div):
        if has_bn:
            w_ok, b = fold_conv(weights[name + "/weights"], None, weights[name + "/BatchNorm/beta"],
                                weights[name + "/BatchNorm/moving_mean"],
                                weights[name + "/BatchNorm/moving_variance"], DECODER_BN_EPS)
        else:
            w_ok, b = fold_conv(weights[name + "/weights"], bias=weights[name + "/biases"])
        if name.startswith("squash"):
            # the 1x1 over concat([crop, full]) runs as two GEMMs over the halves of K (no concat buffer):
            # first half raw, second half adds the first as residual and applies bias + ReLU
            half = cin // 2
            blob.add_layer(w_ok[:, :half], None, half, 1, 1, 1, False)
            blob.add_layer(w_ok[:, half:], b, half, 1, 1, 1, relu)
        else:
            blob.add_layer(w_ok, b, cin, kh, kw, 1, relu)
    return blob.finish()


def _fc(weights, name, kpad=None):
    w = weights[name + "/weights"].T.astype(np.float32)  # (out, in)
    if kpad is not None and kpad != w.shape[1]:
        w = _pad_k(w, kpad)
    return w, weights[name + "/biases"].astype(np.float32)


def round32(v):
    return (v + 31) // 32 * 32


def pack_heads(weights, feat_elems=18432, num_classes=1, num_alpha_bins=12):
    """Seven GEMM layers (include/monopsr_hip.h MPSR_HEAD_LAYERS): the two img_fc layers share their input and are
    fused along N; lwh+alpha and cen_y+cen_z likewise.  Concat inputs are zero-padded to a multiple of 32."""
    p = "output/proposal_fc/proposal_fc/"
    r = "output/regression_fc/regression_fc/"
    blob = Blob()
    wp, bp = _fc(weights, p + "img_fc")
    wr, br = _fc(weights, r + "img_fc")
    blob.add_layer(np.concatenate([wp, wr], 0), np.concatenate([bp, br]), feat_elems, 1, 1, 1, True)
    prop_in = weights[p + "fc0/weights"].shape[0]
    w, b = _fc(weights, p + "fc0", round32(prop_in))
    blob.add_layer(w, b, round32(prop_in), 1, 1, 1, True)
    w, b = _fc(weights, p + "fc1")
    blob.add_layer(w, b, w.shape[1], 1, 1, 1, True)
    wl, bl = _fc(weights, "output/lwh/lwh")
    wa, ba = _fc(weights, "output/alpha")
    blob.add_layer(np.concatenate([wl, wa], 0), np.concatenate([bl, ba]), wl.shape[1], 1, 1, 1, False)
    reg_in = weights[r + "fc0/weights"].shape[0]
    w, b = _fc(weights, r + "fc0", round32(reg_in))
    blob.add_layer(w, b, round32(reg_in), 1, 1, 1, True)
    w, b = _fc(weights, r + "fc1")
    blob.add_layer(w, b, w.shape[1], 1, 1, 1, True)
    wy, by = _fc(weights, "output/cen_y/cen_y")
    wz, bz = _fc(weights, "output/cen_z_offs/cen_z")
    blob.add_layer(np.concatenate([wy, wz], 0), np.concatenate([by, bz]), wy.shape[1], 1, 1, 1, False)
    return blob.finish()


def variable_shapes(scopes=(CROP_SCOPE,), feat_elems=18432, width_div=1):
    """name -> shape of every variable synthetic_weights() would create (no data generated)."""
    shapes = {}
    for scope in scopes:
        for s in scaled_trunk_specs(scope, width_div):
            shapes[s["name"] + "/weights"] = (s["kh"], s["kw"], s["cin"], s["cout"])
            for k in ("gamma", "beta", "moving_mean", "moving_variance"):
                shapes[s["name"] + "/BatchNorm/" + k] = (s["cout"],)
    for name, kh, kw, cin, cout, has_bn, has_bias, _ in scaled_decoder_specs(width_div):
        shapes[name + "/weights"] = (kh, kw, cin, cout)
        if has_bn:
            for k in ("beta", "moving_mean", "moving_variance"):
                shapes[name + "/BatchNorm/" + k] = (cout,)
        if has_bias:
            shapes[name + "/biases"] = (cout,)
    for name, fin, fout, _ in head_fc_specs(feat_elems):
        shapes[name + "/weights"] = (fin, fout)
        shapes[name + "/biases"] = (fout,)
    return shapes
