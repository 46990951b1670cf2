"""us per launch of the xyz-map head's backward (csrc/thin_conv.hip: weight gradient 128 -> 4 channels and data
gradient 4 -> 128 channels on the 48x48 map at batch 256) on the thin-side kernels and on the general ones
(mpsr_debug_set_thin_conv(0)), with the bytes each has to move."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402
from monopsr_amd.core import device_net as dn  # noqa: E402


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


def main():
    lib = _lib.lib()
    B, H, W, C = 256, 48, 48, 128
    x = torch.randn((B, H, W, C), device="cuda")
    dy = torch.randn((B, H, W, 4), device="cuda")
    wd = torch.randn((C, 36), device="cuda")
    dw = torch.zeros((4, 9 * C), device="cuda")
    db = torch.zeros((4,), device="cuda")
    mb = x.numel() * 4 / 1e6
    for thin in (1, 0, 1, 0):
        lib.mpsr_debug_set_thin_conv(thin)
        t1 = timed(lambda: _lib.check(lib.mpsr_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), B, H, W, C, 4, 3, 3, 1,
                                                                _lib.ptr(dw), _lib.ptr(db), _lib.stream())))
        t2 = timed(lambda: dn.conv2d(dy, wd, None, None, 3, 3, 1, False))
        print("thin=%d  weight gradient %7.1f us (%4.2f TB/s of x)   data gradient %7.1f us (%4.2f TB/s of dx)" %
              (thin, t1, mb / t1, t2, mb / t2))
    lib.mpsr_debug_set_thin_conv(1)


if __name__ == "__main__":
    main()
