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


def measure_valu(waves, nv, iters=4000, cus=None, reps=3):
    """MFMA rate with `nv` independent VALU instructions issued after every MFMA of every wave."""
    lib = _lib.lib()
    cus = cus or torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.zeros(4, device="cuda")
    lib.mpsr_debug_mfma_valu.argtypes = [_lib.c_f, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_f]
    _lib.check(lib.mpsr_debug_mfma_valu(out.data_ptr(), cus, waves, nv, 100, _lib.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.mpsr_debug_mfma_valu(out.data_ptr(), cus, waves, nv, iters, _lib.stream()))
    e1.record()
    torch.cuda.synchronize()
    s = e0.elapsed_time(e1) * 1e-3 / reps
    return cus * 4 * waves * iters * 16 * 4096.0 / s / 1e12


def measure_mix(waves, nv, kind, iters=4000, cus=None, reps=3):
    lib = _lib.lib()
    cus = cus or torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.zeros(4, device="cuda")
    lib.mpsr_debug_mfma_mix.argtypes = [_lib.c_f, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_f]
    _lib.check(lib.mpsr_debug_mfma_mix(out.data_ptr(), cus, waves, nv, kind, 100, _lib.stream()))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        _lib.check(lib.mpsr_debug_mfma_mix(out.data_ptr(), cus, waves, nv, kind, iters, _lib.stream()))
    e1.record()
    torch.cuda.synchronize()
    s = e0.elapsed_time(e1) * 1e-3 / reps
    return cus * 4 * waves * iters * 16 * 4096.0 / s / 1e12


def dispatch_rate():
    lib = _lib.lib()
    lib.mpsr_debug_dispatch.argtypes = [_lib.c_f, _lib.c_i, _lib.c_i, _lib.c_i, _lib.c_f]
    for blocks in (2304, 9216, 36864):
        for lds in (0, 18432, 40960):
            for spin in (0, 16, 256):
                for _ in range(2):
                    _lib.check(lib.mpsr_debug_dispatch(None, blocks, lds, spin, _lib.stream()))
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    _lib.check(lib.mpsr_debug_dispatch(None, blocks, lds, spin, _lib.stream()))
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 100.0
                print("%6d workgroups, %5d B LDS, sleep %5d cycles: %7.1f us  (%.1f ns per workgroup)"
                      % (blocks, lds, spin * 64, us, us * 1e3 / blocks))


if __name__ == "__main__":
    if "--dispatch" in sys.argv:
        dispatch_rate()
        sys.exit(0)
    if "--mix" in sys.argv:
        for kind, name in ((0, "v_fma_f32"), (1, "s_add_u32"), (2, "ds_read_b128"), (3, "s_nop")):
            print("%-13s " % name + "  ".join("%dw x%d: %.1f" % (w, nv, measure_mix(w, nv, kind, iters=4000 // w))
                                              for w in (1, 4, 7) for nv in (4, 8)))
        sys.exit(0)
    if "--stores" in sys.argv:
        # LDS stores next to MFMAs: TFLOP/s with n stores after every MFMA of every wave (1, 2 waves per SIMD)
        for kind, name in ((4, "ds_write_b32"), (5, "ds_write_b64"), (6, "ds_write_b128"), (7, "ds_write2st64_b32")):
            print("%-18s " % name + "  ".join("%dw x%d: %.1f" % (w, nv, measure_mix(w, nv, kind, iters=4000 // w))
                                               for w in (1, 2) for nv in (1, 2, 4, 8)))
        sys.exit(0)
    if "--valu" in sys.argv:
        for w in (1, 2, 4, 7):
            print("%d waves/SIMD, VALU per MFMA -> TFLOP/s: " % w + "  ".join(
                "%d: %.1f" % (nv, measure_valu(w, nv, iters=4000 // w)) for nv in (0, 1, 2, 4, 6, 8, 12, 16)))
        sys.exit(0)
    for mode in (0, 1):
        print("lds mode %d: " % mode + "  ".join("%dw %.1f" % (w, measure_lds(w, mode, iters=8000 // w))
                                                 for w in (1, 2, 3, 4, 6, 7, 8)))
    for chains in (4, 1):
        print("chains %d: " % chains + "  ".join("%dw %.1f" % (w, measure(w, chains, iters=8000 // w)) for w in (1, 2, 4, 7, 8)))
