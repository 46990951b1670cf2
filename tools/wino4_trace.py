"""Per-wave timeline of two steady-state K steps of wino4_conv_kernel (library built with -DW4_TRACE):
    tools/build_variant.sh w4trace winograd4.hip -DW4_TRACE
    MPSR_LIB_PATH=abl/w4trace.so python tools/wino4_trace.py [--shape 48,48,256,128]
Stamps per wave (s_memtime cycles): K steps 8 and 9: step start, after each of the three units of 12 MFMAs, after the
barrier; then the end of the K loop.  Waves 0-3 request a patch in the even step and transform it in the odd one,
waves 4-7 the other way round."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="48,48,256,128")
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
H, W, C, N = [int(v) for v in args.shape.split(",")]
B = args.batch
lib = _lib.lib()
lib.mpsr_debug_set_conv_winograd(2)
x = torch.randn((B, H, W, C), device="cuda").clamp_(min=0)
w = torch.randn((N, 9 * C), device="cuda") * 0.05
y = torch.empty((B, H, W, N), device="cuda")
nws = lib.mpsr_conv2d_scratch_floats(B, H, W, N)
ws = torch.empty((nws,), device="cuda")
grid = ((B * (H // 4) * (W // 4) + 31) // 32 + 7) // 8 * 8 * ((N + 63) // 64)
trace = torch.zeros((grid * 8, 16), dtype=torch.int64, device="cuda")
lib.mpsr_debug_set_wino4_trace.argtypes = [ctypes.c_void_p]
lib.mpsr_debug_set_wino4_trace(trace.data_ptr())
for _ in range(3):
    _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, W, C, w.data_ptr(), None, None, y.data_ptr(), N, 3, 3, 1, 1,
                                        0, ws.data_ptr(), nws, _lib.stream()))
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(grid, 8, 16)
t = t[t[:, 0, 0] != 0]
d = np.diff(t, axis=2).astype(np.float64)  # 15 intervals between the 16 stamps
names = ["prologue", "steps 0-7", "step 8: unit 0", "  unit 1", "  unit 2", "  barrier", "step 9: unit 0", "  unit 1",
         "  unit 2", "  barrier", "steps 10..", "(loop exit)", "epilogue: exchange 0", "  finish round 0",
         "  exchange + finish 1"]
print("%d workgroups; %d K steps; cycles, median over workgroups, per wave 0..7" % (len(t), C // 8))
for i, nme in enumerate(names):
    print("  %-22s " % nme + " ".join("%7.0f" % np.median(d[:, wv, i]) for wv in range(8)))
tot = t[:, :, 15].max(axis=1) - t[:, :, 0].min(axis=1)
loop = t[:, :, 12].max(axis=1) - t[:, :, 1].min(axis=1)
print("  workgroup: %.0f cycles; K loop %.0f = %.0f per step (4608 of matrix pipe per SIMD); prologue %.0f; "
      "epilogue %.0f" % (np.median(tot), np.median(loop), np.median(loop) / (C // 8),
                         np.median(t[:, :, 1].min(axis=1) - t[:, :, 0].min(axis=1)),
                         np.median(t[:, :, 15].max(axis=1) - t[:, :, 12].max(axis=1))))
