"""The kernels of the atrous 3x3 layers whose pixel sub-grids are single tiles, side by side (mpsr_debug_set_wino3_form):
0 = F(3x3,3x3), a tile's 25 positions shared by eight waves (csrc/winograd3.hip), 1 = F(3x3,3x3), one wave owns all 25
(csrc/winograd3w.hip: identical bits to 0), 2 = the SIXTEEN-product form of a zero-padded tile, one wave owns all 16
(csrc/winograd3z.hip).

    python tools/wino3_forms.py [--batches 32,64,128,256] [--rounds 5] [--reps 20] [--relu-input]

Checks that both return identical bits (also on a data-gradient launch with a mask through mpsr_conv2d_relu_masked_f32
when --mask is given), then times them interleaved: microseconds and TFLOP/s of EXECUTED multiply-adds (25 per tile and
channel pair).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32,64,128,256")
    ap.add_argument("--shape", default="256,256,4", help="C,N,dilation")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--relu-input", action="store_true", help="post-ReLU-like operands (as in the step)")
    ap.add_argument("--zero-input", action="store_true", help="all-zero activations: the clock the board reaches when "
                    "the matrix pipes toggle nothing (the kernels are power-limited on dense data)")
    ap.add_argument("--sparsity", type=float, default=-1.0, help="fraction of activations set to zero at random")
    args = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda")
    C, N, dil = [int(v) for v in args.shape.split(",")]
    H = 3 * dil
    lib.mpsr_debug_set_conv_winograd(3)
    for B in [int(v) for v in args.batches.split(",")]:
        torch.manual_seed(B)
        x = torch.randn((B, H, H, C), device=dev)
        if args.relu_input:
            x = torch.relu(x)
        if args.zero_input:
            x = torch.zeros_like(x)
        if args.sparsity >= 0:
            x = x * (torch.rand_like(x) >= args.sparsity)
        w = torch.randn((N, 9 * C), device=dev) / (9 * C) ** 0.5
        bias = torch.randn((N,), device=dev)
        nws = lib.mpsr_conv2d_scratch_floats(B, H, H, N)
        ws = torch.empty((nws,), device=dev)
        outs, times = {}, {0: [], 1: [], 2: []}

        def run(y):
            _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, H, C, w.data_ptr(), bias.data_ptr(), None,
                                                y.data_ptr(), N, 3, 3, dil, 1, 0, ws.data_ptr(), nws, _lib.stream()))
        for form in (0, 1, 2):
            lib.mpsr_debug_set_wino3_form(form)
            y = torch.full((B, H, H, N), float("nan"), device=dev)
            run(y)
            torch.cuda.synchronize()
            outs[form] = y
        same = torch.equal(outs[0], outs[1])
        ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(),
                                         w.view(N, 3, 3, C).permute(0, 3, 1, 2).double(), bias.double(),
                                         padding=dil, dilation=dil).relu().permute(0, 2, 3, 1)
        err = [float((outs[f].double() - ref).abs().max() / ref.abs().max().clamp_min(1e-30)) for f in (0, 1, 2)]
        for _ in range(args.rounds):
            for form in (0, 1, 2):
                lib.mpsr_debug_set_wino3_form(form)
                y = outs[form]
                run(y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run(y)
                e1.record()
                torch.cuda.synchronize()
                times[form].append(e0.elapsed_time(e1) * 1e3 / args.reps)
        flop = 2.0 * B * dil * dil * 25 * C * N
        med = {f: sorted(times[f])[len(times[f]) // 2] for f in (0, 1, 2)}
        print("B %4d  identical bits (0 vs 1) %s  err vs fp64 %.2e / %.2e / %.2e | shared positions %7.1f us %6.1f TF/s | one "
              "wave per block %7.1f us %6.1f TF/s | sixteen products %7.1f us (%6.1f TF/s of its own 16-product work)" % (
                  B, same, err[0], err[1], err[2], med[0], flop / med[0] / 1e6, med[1], flop / med[1] / 1e6, med[2],
                  flop * 16 / 25 / med[2] / 1e6), flush=True)
    lib.mpsr_debug_set_wino3_form(-1)
    lib.mpsr_debug_set_conv_winograd(-1)


if __name__ == "__main__":
    main()
