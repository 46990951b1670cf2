"""Attainability reference for the trunk's 1x1 layers: the vendor fp32 GEMM (torch.mm -> rocBLAS / hipBLASLt, TF32
off) on the same (M, N, K) as this library's kernels, in the same process on the same board.

    python tools/gemm_reference.py [--batch 256]

Measurement only: nothing in the package calls a library GEMM.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402

# name, (H, W), C, N, k, dilation
LAYERS = [
    ("b3 conv1 1024->256", (12, 12), 1024, 256, 1, 1),
    ("b3 conv3 256->1024", (12, 12), 256, 1024, 1, 1),
    ("b3 conv2 3x3 256 d4", (12, 12), 256, 256, 3, 4),
    ("squash half 1024->512", (12, 12), 1024, 512, 1, 1),
    ("dec conv2_1 3x3 512->256", (24, 24), 512, 256, 3, 1),
    ("dec conv3_1 3x3 256->128", (48, 48), 256, 128, 3, 1),
]


def timed(f, reps):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    torch.backends.cuda.matmul.allow_tf32 = False
    dev = torch.device("cuda")
    lib = _lib.lib()
    B = args.batch
    print("%-26s %8s %6s %6s | %-22s | %-22s" % ("layer", "M", "N", "K", "vendor GEMM us (TF/s)", "this library us (TF/s)"))
    for name, (H, W), C, N, k, dil in LAYERS:
        M, K = B * H * W, k * k * C
        a = torch.randn(M, K, device=dev)   # the im2col matrix a library GEMM would need (not counted in its time)
        wt = torch.randn(N, K, device=dev) * 0.05
        bias = torch.randn(N, device=dev)
        t_lib = timed(lambda: torch.addmm(bias, a, wt.t()), args.reps)
        del a
        x = torch.randn(B, H, W, C, device=dev)
        y = torch.empty(B, H, W, N, device=dev)
        nws = lib.mpsr_conv2d_scratch_floats(B, H, W, N)
        ws = torch.empty(nws, device=dev)

        def ours():
            _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, W, C, wt.data_ptr(), bias.data_ptr(), None,
                                                y.data_ptr(), N, k, k, dil, 0, 0, ws.data_ptr(), nws, _lib.stream()))
        t_own = timed(ours, args.reps)
        fl = 2.0 * M * N * K
        print("%-26s %8d %6d %6d | %10.1f (%6.1f)    | %10.1f (%6.1f)" % (name, M, N, K, t_lib, fl / t_lib / 1e6,
                                                                         t_own, fl / t_own / 1e6))


if __name__ == "__main__":
    main()
