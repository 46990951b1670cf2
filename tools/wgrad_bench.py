"""Per-layer-shape timing of mpsr_conv2d_wgrad_f32 (weight + bias gradient) on the shapes of one training step.

    python tools/wgrad_bench.py [--batch 256]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402

SHAPES = [  # name, launches per step, H, W, C, N, k, dilation
    ("b1 conv2 3x3 64", 3, 12, 12, 64, 64, 3, 1),
    ("b2 conv2 3x3 128 d2", 4, 12, 12, 128, 128, 3, 2),
    ("b2 conv3 128->512", 4, 12, 12, 128, 512, 1, 1),
    ("b3 conv1 1024->256", 22, 12, 12, 1024, 256, 1, 1),
    ("b3 conv2 3x3 256 d4", 23, 12, 12, 256, 256, 3, 4),
    ("b3 conv3 256->1024", 23, 12, 12, 256, 1024, 1, 1),
    ("squash half 1024->512", 2, 12, 12, 1024, 512, 1, 1),
    ("dec conv2_1 3x3 512->256", 1, 24, 24, 512, 256, 3, 1),
    ("dec conv2_2 3x3 256->256", 1, 24, 24, 256, 256, 3, 1),
    ("dec conv3_1 3x3 256->128", 1, 48, 48, 256, 128, 3, 1),
    ("dec conv3_2 3x3 128->128", 1, 48, 48, 128, 128, 3, 1),
    ("fc 18432->1024", 2, 1, 1, 18432, 1024, 1, 1),
    # the decoder's upsampled layers in tap-GEMM form: a 1x1 weight gradient on the SOURCE map with 9 N outputs
    ("dec conv2_1 tap GEMM 512->9x256", 1, 12, 12, 512, 2304, 1, 1),
    ("dec conv3_1 tap GEMM 256->9x128", 1, 24, 24, 256, 1152, 1, 1),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--target", type=int, default=0, help="debug: workgroups per launch the pixel slicing aims at")
    ap.add_argument("--min-steps", type=int, default=-1, help="debug: 32-pixel steps per slice of a 1x1 layer at least "
                                                               "(default: the library's 12; 0 = the plain target)")
    ap.add_argument("--ungrouped", action="store_true", help="A/B: the workgroup order that ignores the XCDs")
    ap.add_argument("--direct", action="store_true", help="A/B: 1x1 layers on pw_wgrad_direct_kernel (operands straight "
                                                          "into the MFMA's source registers) instead of the LDS-staged kernel")
    args = ap.parse_args()
    B = args.batch
    lib = _lib.lib()
    if args.target:
        lib.mpsr_debug_set_wgrad_target(args.target)
    lib.mpsr_debug_set_wgrad_direct(1 if args.direct else 0)
    if args.min_steps >= 0:
        lib.mpsr_debug_set_wgrad_min_steps(args.min_steps)
    if args.ungrouped:
        lib.mpsr_debug_set_wgrad_grouped(0)
    total = 0.0
    print("%-28s %3s %9s %10s %8s" % ("layer", "n", "GFLOP", "us", "TF/s"))
    for name, count, H, W, C, N, k, dil in SHAPES:
        x = torch.randn((B, H, W, C), device="cuda")
        dy = torch.randn((B, H, W, N), device="cuda")
        dw = torch.zeros((N, k * k * C), device="cuda")
        db = torch.zeros((N,), device="cuda")
        s = _lib.stream()

        def run():
            _lib.check(lib.mpsr_conv2d_wgrad_f32(_lib.ptr(x), _lib.ptr(dy), B, H, W, C, N, k, k, dil, _lib.ptr(dw),
                                                 _lib.ptr(db), s))
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        gf = 2.0 * B * H * W * C * N * k * k / 1e9
        total += us * count
        print("%-28s %3d %9.2f %10.1f %8.1f" % (name, count, gf, us, gf / us * 1e3))
    print("per-step wgrad time (ms): %.2f" % (total / 1e3))


if __name__ == "__main__":
    main()
