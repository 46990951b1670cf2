"""Per-wave timeline of one steady-state tile (the workgroup's third) of pw_conv_kernel (library built with
-DPWS_TRACE):
    tools/build_variant.sh pwstrace pointwise.hip -DPWS_TRACE
    MPSR_LIB_PATH=abl/pwstrace.so python tools/pws_trace.py [--shape 256,1024,1]
shape = K,N,residual on the 12x12 trunk map.  Cycles between stamps, median over workgroups, per wave."""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="256,1024,1")
ap.add_argument("--batch", type=int, default=256)
args = ap.parse_args()
K, N, res = [int(v) for v in args.shape.split(",")]
B, H, W = args.batch, 12, 12
lib = _lib.lib()
lib.mpsr_debug_set_conv_pointwise(1)
x = torch.randn((B, H, W, K), device="cuda").clamp_(min=0)
w = torch.randn((N, K), device="cuda") * 0.05
r = torch.randn((B, H, W, N), device="cuda") if res else None
y = torch.empty((B, H, W, N), device="cuda")
grid = 1024
trace = torch.zeros((grid * 4, 16), dtype=torch.int64, device="cuda")
lib.mpsr_debug_set_pointwise_trace.argtypes = [ctypes.c_void_p]
lib.mpsr_debug_set_pointwise_trace(trace.data_ptr())
for _ in range(3):
    _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, W, K, w.data_ptr(), None, r.data_ptr() if res else None,
                                        y.data_ptr(), N, 1, 1, 1, 1, 0, None, 0, _lib.stream()))
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(grid, 4, 16)
t = t[t[:, 0, 0] != 0]
names = ["K loop of the tile"]
d = np.diff(t[:, :, :2], axis=2).astype(np.float64)
print("%d workgroups; K = %d: %d stages of 48 MFMAs (3072 cycles of matrix pipe each)" % (len(t), K, K // 32))
for i, nme in enumerate(names):
    print("  %-36s " % nme + " ".join("%7.0f" % np.median(d[:, wv, i]) for wv in range(4)))
print("  tile: %.0f cycles (median over workgroups, wave 0); matrix pipe alone would need %d" % (
    np.median(t[:, 0, 1] - t[:, 0, 0]), (K // 32) * 3072))
