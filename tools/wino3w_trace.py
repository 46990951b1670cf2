"""Per-wave cycle stamps of the one-wave-per-tile-block kernels of block3's atrous layers (need a trace build:
tools/build_variant.sh w3wtrace winograd3w.hip -DW3W_TRACE -fno-slp-vectorize   (F(3x3,3x3), 25 positions, --form 1)
tools/build_variant.sh w3ztrace winograd3z.hip -DW3Z_TRACE -fno-slp-vectorize   (sixteen-product form, --form 2)).

    MPSR_LIB_PATH=abl/w3ztrace.so python tools/wino3w_trace.py --form 2 [--batch 256]

Prints, for the waves of the first workgroups: prologue, every K step, the wait states behind the loop, epilogue, in
shader cycles, next to the matrix pipe's own time for a step (100 MFMAs x 64 cycles).
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--form", type=int, default=1, choices=[1, 2])
    args = ap.parse_args()
    lib = _lib.lib()
    dev = torch.device("cuda")
    B, C, N, dil = args.batch, 256, 256, 4
    H = 3 * dil
    x = torch.relu(torch.randn((B, H, H, C), device=dev))
    w = torch.randn((N, 9 * C), device=dev) / (9 * C) ** 0.5
    bias = torch.randn((N,), device=dev)
    y = torch.empty((B, H, H, N), device=dev)
    nws = lib.mpsr_conv2d_scratch_floats(B, H, H, N)
    ws = torch.empty((nws,), device=dev)
    lib.mpsr_debug_set_conv_winograd(3)
    lib.mpsr_debug_set_wino3_form(args.form)
    step_cycles = 6400 if args.form == 1 else 4096  # MFMAs of a K step x 64
    def run():
        _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, H, C, w.data_ptr(), bias.data_ptr(), None, y.data_ptr(),
                                            N, 3, 3, dil, 1, 0, ws.data_ptr(), nws, _lib.stream()))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    launch_us = e0.elapsed_time(e1) * 1e3 / 20
    n = 8 * 4 * 40
    buf = (ctypes.c_ulonglong * n)()
    fn = lib.mpsr_debug_wino3w_trace if args.form == 1 else lib.mpsr_debug_wino3z_trace
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
    assert fn(buf, n) == 0
    print("matrix pipe per K step: %d cycles; per wave and launch: %d" % (step_cycles, step_cycles * (C // 8)))
    life = [buf[(b * 4 + v) * 40 + 35] - buf[(b * 4 + v) * 40] for b in range(8) for v in range(4)
            if buf[(b * 4 + v) * 40 + 35]]
    if life:
        med = sorted(life)[len(life) // 2]
        print("launch %.1f us (back to back, events); a wave lives %d shader cycles (s_memtime) -> shader clock >= %.2f GHz "
              "under this kernel; MFMA issue share of a wave's cycles %.3f, of the launch's time at 2.4 GHz %.3f" % (
                  launch_us, med, med / launch_us / 1e3, float(step_cycles) * (C // 8) / med, float(step_cycles) * (C // 8) / (launch_us * 2400.0)))
    for blk in range(8):
        for wv in range(4):
            t = buf[(blk * 4 + wv) * 40:(blk * 4 + wv + 1) * 40]
            if t[35] == 0:
                continue
            steps = [t[2 + s] - t[1 + s] for s in range(C // 8)]
            print("wg %d wave %d: prologue %6d | K loop %7d (steps min %5d med %5d max %5d; first %5d last %5d) | "
                  "drain %4d | epilogue %6d | total %7d" % (
                      blk, wv, t[1] - t[0], t[1 + C // 8] - t[1], min(steps), sorted(steps)[len(steps) // 2], max(steps),
                      steps[0], steps[-1], t[34] - t[1 + C // 8], t[35] - t[34], t[35] - t[0]))
    lib.mpsr_debug_set_wino3_form(-1)
    lib.mpsr_debug_set_conv_winograd(-1)


if __name__ == "__main__":
    main()
