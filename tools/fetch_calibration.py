"""Streams a 1 GiB buffer once with each access pattern of mpsr_debug_fetch_calibration (run under
rocprofv3 --pmc FETCH_SIZE by tools/fetch_calibration.sh): what FETCH_SIZE reports per byte really read."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402

lib = _lib.lib()
fn = lib.mpsr_debug_fetch_calibration
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
fn.restype = ctypes.c_int
n = 1 << 28  # floats = 1 GiB: four times the 256 MiB Infinity Cache
buf = torch.randn((n,), device="cuda")
sink = torch.zeros((4,), device="cuda")
for mode in (0, 1, 0, 1):
    _lib.check(fn(buf.data_ptr(), n, mode, sink.data_ptr(), _lib.stream()))
torch.cuda.synchronize()
print("bytes per launch: %d" % (4 * n))
