"""fp32 MFMA rate this box sustains (v_mfma_f32_32x32x2_f32 only), by waves per SIMD and accumulator chains per wave.

    python tools/mfma_peak.py
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from monopsr_amd import _lib  # noqa: E402


def measure(waves, chains, iters=4000, cus=None, reps=3):
    lib = _lib.lib()
    cus = cus or torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.zeros(4, device="cuda")
    lib.mpsr_debug_mfma_peak.argtypes = [_lib.c_f, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_f]
    _lib.check(lib.mpsr_debug_mfma_peak(out.data_ptr(), cus, waves, chains, 100, _lib.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.mpsr_debug_mfma_peak(out.data_ptr(), cus, waves, chains, iters, _lib.stream()))
    e1.record()
    torch.cuda.synchronize()
    s = e0.elapsed_time(e1) * 1e-3 / reps
    return cus * 4 * waves * iters * 16 * 4096.0 / s / 1e12


def measure_lds(waves, mode, iters=4000, cus=None, reps=3):
    lib = _lib.lib()
    cus = cus or torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.zeros(4, device="cuda")
    lib.mpsr_debug_mfma_lds.argtypes = [_lib.c_f, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_f]
    _lib.check(lib.mpsr_debug_mfma_lds(out.data_ptr(), cus, waves, mode, 100, _lib.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.mpsr_debug_mfma_lds(out.data_ptr(), cus, waves, mode, iters, _lib.stream()))
    e1.record()
    torch.cuda.synchronize()
    s = e0.elapsed_time(e1) * 1e-3 / reps
    return cus * 4 * waves * iters * 16 * 4096.0 / s / 1e12


if __name__ == "__main__":
    for mode in (0, 1):
        print("lds mode %d: " % mode + "  ".join("%dw %.1f" % (w, measure_lds(w, mode, iters=8000 // w))
                                                 for w in (1, 2, 3, 4, 6, 7, 8)))
    for chains in (4, 1):
        print("chains %d: " % chains + "  ".join("%dw %.1f" % (w, measure(w, chains, iters=8000 // w)) for w in (1, 2, 4, 7, 8)))
