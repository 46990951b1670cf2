"""Phase timeline of wino_conv_kernel workgroups (library built with -DWINO_TRACE):
    MPSR_LIB_PATH=abl/wtrace.so python tools/wino_trace.py [--shape 48,48,256,128]"""
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
lib.mpsr_debug_set_conv_winograd(1)
lib.mpsr_debug_set_wino_waves(4)  # the stamps are in the four-wave kernel
x = torch.randn((B, H, W, C), device="cuda")
w = torch.randn((N, 9 * C), device="cuda") * 0.05
y = torch.empty((B, H, W, N), device="cuda")
nws = lib.mpsr_conv2d_scratch_floats(B, H, W, N)
ws = torch.empty((nws,), device="cuda")
trace = torch.zeros((1 << 17, 8), dtype=torch.int64, device="cuda")
lib.mpsr_debug_set_wino_trace.argtypes = [ctypes.c_void_p]
lib.mpsr_debug_set_wino_trace(trace.data_ptr())
for _ in range(3):
    _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, W, C, w.data_ptr(), None, None, y.data_ptr(), N, 3, 3, 1, 1,
                                        0, ws.data_ptr(), nws, _lib.stream()))
torch.cuda.synchronize()
t_all = trace.cpu().numpy()
nwg = int((t_all[:, 0] != 0).sum()) // 2 if (t_all[:, 2] != 0).sum() < (t_all[:, 0] != 0).sum() else int((t_all[:, 0] != 0).sum())
grid = ((B * (H // 2) * (W // 2) + 63) // 64 + 7) // 8 * 8 * ((N + 63) // 64)
sums = t_all[grid:2 * grid]
sums = sums[sums[:, 0] != 0]
t = t_all[:grid]
t = t[t[:, 0] != 0]
if len(sums):
    it = C // 16 - 1
    print("steady state per half (median over workgroups): second halves (g1) %.0f  first halves (g0) %.0f cycles"
          % (np.median(sums[:, 0]) / it, np.median(sums[:, 1]) / it))
d = np.diff(t, axis=1).astype(np.float64)
names = ["prologue", "half0 (no stores)", "half1(0)+stores", "half0(1)+stores", "rest of loop", "last half", "epilogue"]
print("%d workgroups; cycles (s_memtime units), median over workgroups; K steps = %d" % (len(t), C // 16))
for i, nme in enumerate(names):
    print("  %-20s %9.0f" % (nme, np.median(d[:, i])))
print("  total               %9.0f" % np.median(t[:, 7] - t[:, 0]))
