"""Per-workgroup timeline of pw_conv_kernel (library built with -DPW_TRACE):
    tools/build_variant.sh pwtrace pointwise.hip -DPW_TRACE
    MPSR_LIB_PATH=abl/pwtrace.so python tools/pw_trace.py [--shape 256,1024,1]
shape = K,N,residual on the 12x12 trunk map.  Prints the median phase lengths (cycles) and, from the 100 MHz real-time
counter, how the two workgroups of a CU overlap."""
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
M = B * H * W
grid = ((M + 287) // 288 + 7) // 8 * 8 * ((N + 127) // 128)
trace = torch.zeros((grid * 4, 16), dtype=torch.int64, device="cuda")
lib.mpsr_debug_set_pointwise_trace.argtypes = [ctypes.c_void_p]
lib.mpsr_debug_set_pointwise_trace(trace.data_ptr())
for _ in range(3):
    _lib.check(lib.mpsr_conv2d_nhwc_f32(x.data_ptr(), B, H, W, K, w.data_ptr(), None, r.data_ptr() if res else None,
                                        y.data_ptr(), N, 1, 1, 1, 1, 0, None, 0, _lib.stream()))
torch.cuda.synchronize()
t = trace.cpu().numpy().reshape(grid, 4, 16)
t = t[t[:, 0, 0] != 0]
n = int((t[0, 0, :12] != 0).sum())
d = np.diff(t[:, :, :n], axis=2).astype(np.float64)
print("%d workgroups, %d stamps; cycles, median / p90 over workgroups (wave 0)" % (len(t), n))
labels = ["prologue"] + ["K loop part %d" % i for i in range(n - 4)] + ["exchange + stores issued", "stores acknowledged"]
for i in range(n - 1):
    print("  %-22s %8.0f %8.0f" % (labels[i], np.median(d[:, 0, i]), np.percentile(d[:, 0, i], 90)))
rt0, rt1 = t[:, 0, 12], t[:, :, 13].max(axis=1)
base = rt0.min()
print("  kernel: %.1f us; workgroup lifetime median %.1f us" % ((rt1.max() - base) / 100.0, np.median(rt1 - rt0) / 100.0))
key = (t[:, 0, 15] & 7) * 256 + ((t[:, 0, 14] >> 8) & 0xff)
cus = np.unique(key)
print("  %d distinct CU keys; workgroups per CU: min %d max %d" % (len(cus), min((key == c).sum() for c in cus),
                                                                   max((key == c).sum() for c in cus)))
for c in cus[:3]:
    idx = np.where(key == c)[0]
    idx = idx[np.argsort(rt0[idx])]
    print("  CU %04x: " % c + "  ".join("[%.1f-%.1f]" % ((rt0[i] - base) / 100.0, (rt1[i] - base) / 100.0) for i in idx))
    c0 = t[idx, 0, 0].min()
    for i in idx:  # shader-clock stamps of wave 0, kilocycles from the CU's first start
        print("      start %6.1f  K loop %6.1f .. %6.1f  stores issued %6.1f  done %6.1f" % tuple(
            (t[i, 0, j] - c0) / 1e3 for j in (0, 1, n - 3, n - 2, n - 1)))
