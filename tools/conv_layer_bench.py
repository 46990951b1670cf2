"""Per-layer-shape timing of mpsr_conv2d_nhwc_f32 on the GPU for every tile configuration.

    python tools/conv_layer_bench.py [--batch 256] [--tiles 0,1,2,3]

Prints, per distinct layer shape of the instance path, launches per step, GFLOP, and TFLOP/s for each tile
configuration (the knob is mpsr_debug_set_conv_tile; -1 = the library's heuristic).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402

# (name, count per step, M spatial (H, W), C, N, k, dilation, residual)
SHAPES = [
    ("root 1x1 over im2col", 1, (576, 1), 160, 64, 1, 1, False),
    ("b1 conv1 256->64", 2, (12, 12), 256, 64, 1, 1, False),
    ("b1 conv2 3x3 64", 3, (12, 12), 64, 64, 3, 1, False),
    ("b1 conv3 64->256 +res", 3, (12, 12), 64, 256, 1, 1, True),
    ("b2 conv1 512->128", 3, (12, 12), 512, 128, 1, 1, False),
    ("b2 conv2 3x3 128 d2", 4, (12, 12), 128, 128, 3, 2, False),
    ("b2 conv3 128->512 +res", 4, (12, 12), 128, 512, 1, 1, True),
    ("b3 shortcut 512->1024", 1, (12, 12), 512, 1024, 1, 1, False),
    ("b3 conv1 1024->256", 22, (12, 12), 1024, 256, 1, 1, False),
    ("b3 conv2 3x3 256 d4", 23, (12, 12), 256, 256, 3, 4, False),
    ("b3 conv3 256->1024 +res", 23, (12, 12), 256, 1024, 1, 1, True),
    ("squash half 1024->512", 2, (12, 12), 1024, 512, 1, 1, False),
    ("dec conv2_1 3x3 512->256", 1, (24, 24), 512, 256, 3, 1, False),
    ("dec conv2_2 3x3 256->256", 1, (24, 24), 256, 256, 3, 1, False),
    ("dec conv3_1 3x3 256->128", 1, (48, 48), 256, 128, 3, 1, False),
    ("dec conv3_2 3x3 128->128", 1, (48, 48), 128, 128, 3, 1, False),
    ("xyz 3x3 128->3", 1, (48, 48), 128, 3, 3, 1, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tiles", default="-1,0,3,5")
    ap.add_argument("--classes", type=int, default=-1)
    ap.add_argument("--math", default="fp32", choices=["fp32", "bf16x3"])
    ap.add_argument("--split", type=int, default=1, help="split_k passed to the library (0 = automatic)")
    ap.add_argument("--sched", default="-1", help="comma list of schedules to time per tile: -1 heuristic, 0 one tile "
                    "per workgroup, 1 stream-K; 1:G or 1:-N pins the workgroups per CU / the workgroup count")
    ap.add_argument("--depth", default="-1", help="comma list: register staging depth of the one-tile-per-workgroup "
                    "kernel: -1 heuristic, 1, 2")
    ap.add_argument("--wino", default="-1", help="comma list: Winograd for eligible 3x3 layers: -1 heuristic, 0 never, 1 F(2x2,3x3), 2 F(4x4,3x3)")
    ap.add_argument("--pw", default="-1", help="comma list: the persistent pointwise kernel for 1x1 layers: -1 "
                    "heuristic, 0 never, 1 wherever it applies")
    ap.add_argument("--pws-per-cu", type=int, default=2)
    ap.add_argument("--plain", default="-1", help="comma list: the decode-free 1x1 instantiation: -1 whenever it "
                    "applies, 0 never")
    ap.add_argument("--rounds", type=int, default=1, help="interleaved rounds over all variants of a shape; the "
                    "median over rounds is reported (boxes and power states drift: compare within one run only)")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--shape", action="append", default=[],
                    help="extra shape H,W,C,N,k,dil,res (repeatable); replaces the built-in list")
    args = ap.parse_args()
    _lib.set_conv_math(args.math)
    tiles = [int(t) for t in args.tiles.split(",")]
    lib = _lib.lib()
    lib.mpsr_debug_set_conv_classes(args.classes)
    lib.mpsr_debug_set_pointwise_per_cu(args.pws_per_cu)
    depths = [int(d) for d in args.depth.split(",")]
    dev = torch.device("cuda")
    B = args.batch
    scheds = []
    for sp in args.sched.split(","):
        a, _, b = sp.partition(":")
        scheds.append((int(a), int(b) if b else 0))
    winos = [int(v) for v in args.wino.split(",")]
    plains = [int(v) for v in args.plain.split(",")]
    pws = [int(v) for v in args.pw.split(",")]
    combos = [(t, sc, d, wv, pl, pw) for t in tiles for sc in scheds for d in depths for wv in winos for pl in plains
              for pw in pws]

    def label(c):
        t, (a, b), d, wv, pl, pw = c
        return "t%d s%d:%d d%d w%d p%d pw%d" % (t, a, b, d, wv, pl, pw)
    print("%-28s %3s %9s | %s" % ("layer", "n", "GFLOP", "  ".join(label(c) + " TF/s (us)" for c in combos)))
    total = {c: 0.0 for c in combos}
    shapes = SHAPES
    if args.shape:
        shapes = []
        for sp in args.shape:
            H, W, C, N, k, dil, res = [int(v) for v in sp.split(",")]
            shapes.append(("custom %s" % sp, 1, (H, W), C, N, k, dil, bool(res)))
    for name, count, (H, W), C, N, k, dil, res in shapes:
        if args.only and args.only not in name:
            continue
        x = torch.randn((B, H, W, C), device=dev)
        w = torch.randn((N, k * k * C), device=dev) * 0.05
        bias = torch.randn((N,), device=dev)
        r = torch.randn((B, H, W, N), device=dev) if res else None
        y = torch.empty((B, H, W, N), device=dev)
        nws = lib.mpsr_conv2d_scratch_floats(B, H, W, N) if args.split == 0 else (
            args.split * B * H * W * N if args.split > 1 else 0)
        ws = torch.empty((nws,), device=dev) if nws else None
        flop = 2.0 * B * H * W * C * k * k * N
        cells = []
        samples = {c: [] for c in combos}
        for rnd in range(args.rounds):
            for c in combos:
                t, sc, d, wv, pl, pw = c
                if t == 4 and N > 32:
                    continue
                lib.mpsr_debug_set_conv_tile(t)
                lib.mpsr_debug_set_conv_sched(sc[0], sc[1])
                lib.mpsr_debug_set_conv_depth(d)
                lib.mpsr_debug_set_conv_winograd(wv)
                lib.mpsr_debug_set_conv_plain(pl)
                lib.mpsr_debug_set_conv_pointwise(pw)

                def run():
                    _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, W, C, w.data_ptr(), bias.data_ptr(),
                                                        r.data_ptr() if res else None, y.data_ptr(), N, k, k, dil, 1,
                                                        args.split, ws.data_ptr() if nws else None, nws, _lib.stream()))
                run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.reps):
                    run()
                e1.record()
                torch.cuda.synchronize()
                samples[c].append(e0.elapsed_time(e1) * 1e3 / args.reps)
        for c in combos:
            if not samples[c]:
                cells.append("      -        ")
                continue
            us = sorted(samples[c])[len(samples[c]) // 2]
            total[c] += us * count
            cells.append("%6.1f (%7.1f)" % (flop / us / 1e6, us))
        lib.mpsr_debug_set_conv_tile(-1)
        lib.mpsr_debug_set_conv_sched(-1, 0)
        lib.mpsr_debug_set_conv_depth(-1)
        lib.mpsr_debug_set_conv_winograd(-1)
        lib.mpsr_debug_set_conv_plain(-1)
        lib.mpsr_debug_set_conv_pointwise(-1)
        print("%-28s %3d %9.2f | %s" % (name, count, flop / 1e9, "  ".join(cells)))
    print("per-step conv time (ms): " + "  ".join("%s %.2f" % (label(c), total[c] / 1e3) for c in combos))


if __name__ == "__main__":
    main()
