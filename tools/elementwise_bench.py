"""Timing of the backward pass's elementwise kernels (mpsr_act_bias_grad) on the layer shapes of one training step.

    python tools/elementwise_bench.py [--batch 256]
Prints per shape: microseconds and GB/s of algorithmic traffic (read dy [+ y], write dx) with and without the bias
column sums.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    args = ap.parse_args()
    B = args.batch
    lib = _lib.lib()
    shapes = [("b3 conv3 out 12x12x1024", B * 144, 1024), ("b3 conv1/2 out 12x12x256", B * 144, 256),
              ("b2 out 12x12x512", B * 144, 512), ("dec 24x24x256", B * 576, 256), ("dec 48x48x128", B * 2304, 128),
              ("fc 1024", B, 1024)]
    print("%-28s %10s %10s | %10s %10s | %10s" % ("shape", "relu+db us", "GB/s", "relu us", "GB/s", "db only us"))
    for name, M, N in shapes:
        dy = torch.randn((M, N), device="cuda")
        y = torch.randn((M, N), device="cuda")
        dx = torch.empty_like(dy)
        db = torch.zeros((N,), device="cuda")
        s = _lib.stream()
        full = timed(lambda: _lib.check(lib.mpsr_act_bias_grad(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dx), _lib.ptr(db), M, N, s)))
        relu = timed(lambda: _lib.check(lib.mpsr_act_bias_grad(_lib.ptr(dy), _lib.ptr(y), _lib.ptr(dx), None, M, N, s)))
        dbo = timed(lambda: _lib.check(lib.mpsr_act_bias_grad(_lib.ptr(dy), None, None, _lib.ptr(db), M, N, s)))
        byt = 3.0 * M * N * 4
        print("%-28s %10.1f %10.0f | %10.1f %10.0f | %10.1f" % (name, full, byt / full / 1e3, relu, byt / relu / 1e3, dbo))


if __name__ == "__main__":
    main()
