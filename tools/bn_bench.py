"""us per launch of the training-mode BatchNorm kernels (csrc/batchnorm.hip) on the map decoder's four layers at
batch 256 -- (M, C) = (147456, 256) twice and (589824, 128) twice -- with the bytes each launch has to move."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
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
    s = _lib.stream()
    for M, C in ((147456, 256), (589824, 128)):
        z = torch.randn((M, C), device="cuda")
        dy = torch.randn((M, C), device="cuda")
        y = torch.relu(z)
        dz = torch.empty_like(z)
        mean = torch.zeros((C,), device="cuda")
        istd = torch.ones((C,), device="cuda")
        d0 = torch.zeros((C,), dtype=torch.float64, device="cuda")
        d1 = torch.zeros((C,), dtype=torch.float64, device="cuda")
        mb = M * C * 4 / 1e6
        t = timed(lambda: _lib.check(lib.mpsr_batch_norm_stats(_lib.ptr(z), M, C, _lib.ptr(d0), _lib.ptr(d1), s)))
        print("M %7d C %4d  stats      %7.1f us  %5.2f TB/s" % (M, C, t, mb / t))
        t = timed(lambda: _lib.check(lib.mpsr_batch_norm_apply(_lib.ptr(z), M, C, _lib.ptr(mean), _lib.ptr(istd),
                                                               _lib.ptr(mean), 1, _lib.ptr(dz), s)))
        print("M %7d C %4d  apply      %7.1f us  %5.2f TB/s" % (M, C, t, 2 * mb / t))
        t = timed(lambda: _lib.check(lib.mpsr_batch_norm_grad_sums(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(z), M, C,
                                                                   _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(d0),
                                                                   _lib.ptr(d1), s)))
        print("M %7d C %4d  grad sums  %7.1f us  %5.2f TB/s" % (M, C, t, 3 * mb / t))
        t = timed(lambda: _lib.check(lib.mpsr_batch_norm_grad(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(z), M, C,
                                                              _lib.ptr(mean), _lib.ptr(istd), _lib.ptr(mean),
                                                              _lib.ptr(mean), _lib.ptr(dz), s)))
        print("M %7d C %4d  grad       %7.1f us  %5.2f TB/s" % (M, C, t, 4 * mb / t))


if __name__ == "__main__":
    main()
