"""Randomised sweep of mpsr_conv2d_nhwc_f32 against a float64 reference: small ragged shapes through every schedule the
entry point can pick (tile instantiations, staging depths, layer-kind instantiations, border classes, split-K,
automatic scheduling with scratch, Winograd, stream-K).  Seeded: the same 160 cases every run."""
import numpy as np
import pytest
import torch

from oracle import net as onet

pytestmark = pytest.mark.gpu


def _ref(x, w_hwio, bias, residual, rate, relu):
    y = onet.tf_conv2d(torch.from_numpy(x).double(), torch.from_numpy(w_hwio).double(), rate=rate)
    if bias is not None:
        y = y + torch.from_numpy(bias).double()
    if residual is not None:
        y = y + torch.from_numpy(residual).double()
    return (torch.relu(y) if relu else y).numpy()


@pytest.mark.parametrize("chunk", range(8))
def test_random_layers(chunk):
    from monopsr_amd import _lib
    from monopsr_amd.core import device_net as dn
    from monopsr_amd.core import weights as W
    lib = _lib.lib()
    rng = np.random.default_rng(1000 + chunk)
    dev = torch.device("cuda")
    try:
        for case in range(20):
            k = int(rng.choice([1, 3]))
            dil = int(rng.choice([1, 2, 4])) if k == 3 else 1
            B, H, Wd = int(rng.integers(1, 6)), int(rng.integers(1, 15)), int(rng.integers(1, 15))
            C = int(rng.choice([4, 8, 12, 32, 36, 64, 96]))
            N = int(rng.choice([1, 3, 4, 8, 31, 32, 36, 64, 70, 130]))
            has_bias, has_res, relu = bool(rng.integers(2)), bool(rng.integers(2)), bool(rng.integers(2))
            tile = int(rng.choice([-1, -1, 0, 1, 2, 3, 4, 5]))
            if tile == 4 and N > 32:
                tile = 3
            depth = int(rng.choice([-1, 1, 2]))
            classes = int(rng.choice([-1, 0, 1]))
            split = int(rng.choice([0, 0, 1, 1, 2, 3]))
            plain = int(rng.choice([-1, -1, 0]))
            x = rng.standard_normal((B, H, Wd, C)).astype(np.float32)
            w = (rng.standard_normal((k, k, C, N)) / np.sqrt(k * k * C)).astype(np.float32)
            bias = rng.standard_normal(N).astype(np.float32) if has_bias else None
            res = rng.standard_normal((B, H, Wd, N)).astype(np.float32) if has_res else None
            ref = _ref(x, w, bias, res, dil, relu)
            w_ok, _ = W.fold_conv(w)
            lib.mpsr_debug_set_conv_tile(tile)
            lib.mpsr_debug_set_conv_depth(depth)
            lib.mpsr_debug_set_conv_classes(classes)
            lib.mpsr_debug_set_conv_plain(plain)
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
            got = dn.conv2d(t(x), t(w_ok), t(bias) if has_bias else None, t(res) if has_res else None, k, k, dil, relu,
                            split_k=split).cpu().numpy().astype(np.float64)
            scale = max(1.0, float(np.abs(ref).max()))
            err = float(np.abs(got - ref).max()) / scale
            assert got.shape == ref.shape and err <= 2e-5, \
                "case %d/%d: B%d H%d W%d C%d N%d k%d d%d bias%d res%d relu%d tile%d depth%d classes%d split%d plain%d: " \
                "err %.2e" % (chunk, case, B, H, Wd, C, N, k, dil, has_bias, has_res, relu, tile, depth, classes, split,
                              plain, err)
    finally:
        lib.mpsr_debug_set_conv_tile(-1)
        lib.mpsr_debug_set_conv_depth(-1)
        lib.mpsr_debug_set_conv_classes(-1)
        lib.mpsr_debug_set_conv_plain(-1)
